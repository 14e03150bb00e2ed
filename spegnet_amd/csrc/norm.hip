// LayerNorm (Hiera trunk, eps 1e-6) and BatchNorm2d-on-NHWC (head, eps 1e-5, momentum 0.1) kernels.
// All are HBM-bound: 16-byte vector accesses, one pass over the data per kernel, f32 statistics,
// wave shuffles for the row/channel reductions.
#include "common.h"

namespace spg {

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (<= MAXCH 16-byte chunks per lane).
// ---------------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 5;

// RW = rows per wave: with many short rows (stages 1 and 2: 73728 x 144, 18432 x 288 at batch 8) one row per wave keeps only 288-576 bytes
// in flight per wave -- a launch bound by memory latency at a third of the HBM rate.  A wave then takes RW consecutive rows, requests all of
// them before the first reduction, and works through them one after the other (the arithmetic per row is unchanged: same bits).
// HALF: a row on 32 lanes, two rows per wave -- for rows of at most 32 chunks (stage 1: C = 144 is 18 chunks: 18 of 64 lanes busy with a
// row per wave, 36 of 64 with two).  The row sums stay inside the half (xor distances < 32).
template <int LPR> __device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T, int NI = LN_MAXCH, int RW = 1, bool HALF = false>   // NI = 16-byte chunk slots per lane (ceil(C / (64 VEC))): sized per launch, not for the widest row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                            int C, float eps) {
  constexpr int VEC = ST<T>::VEC;
  static_assert(!HALF || NI == 1, "two rows per wave: one chunk slot per lane");
  constexpr int LPR = HALF ? 32 : 64;
  const int lane = threadIdx.x & (LPR - 1);
  const int row0 = (blockIdx.x * (256 / LPR) + (threadIdx.x / LPR)) * RW;
  if (row0 >= M) return;
  const int nch = C / VEC;
  // Every load of the wave is REQUESTED before the first one is used: written as `if (in range) unpack(load(..))` per chunk slot, each
  // slot's load was waited for inside its own branch -- with two slots (C = 576: 64 + 8 chunks) two dependent memory round trips for the row,
  // and the parameters a third after the reductions (found in the ISA: global_load / s_waitcnt vmcnt(0) pairs).  Out-of-range slots and
  // rows read a valid address (chunk 0 / the wave's first row) and are ignored.
  u32x4 raw[RW][NI];
  f32x4 gq[NI][2 * (VEC / 4)];        // gamma | beta of this lane's chunks (f32: VEC / 4 quads each)
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      const bool ok = ch < nch && row0 + r < M;
      raw[r][i] = ld16(x + (long)(ok ? row0 + r : row0) * C + (ok ? ch : 0) * VEC);
    }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ch = lane + LPR * i;
    const int cc = (ch < nch ? ch : 0) * VEC;
#pragma unroll
    for (int q4 = 0; q4 < VEC / 4; ++q4) {
      gq[i][q4] = *reinterpret_cast<const f32x4*>(gamma + cc + 4 * q4);
      gq[i][VEC / 4 + q4] = *reinterpret_cast<const f32x4*>(beta + cc + 4 * q4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise pulls slot 0's conversion -- and its wait -- ahead of the other requests)
  float v[RW][NI][VEC];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < NI; ++i) unpack16<T>(raw[r][i], v[r][i]);
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[r][i][e];
      }
    }
    const float mu = row_sum<LPR>(s) / C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v[r][i][e] - mu; ss += d * d; }
      }
    }
    const float rs = rsqrtf(row_sum<LPR>(ss) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      if (ch < nch) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (v[r][i][e] - mu) * rs * gq[i][e / 4][e % 4] + gq[i][VEC / 4 + e / 4][e % 4];
        st16(y + (long)row * C + ch * VEC, pack16<T>(o));
      }
    }
  }
}

// (Round 2, measured and dropped: requesting gamma / beta / dres together with the row as 16-byte loads -- one memory round trip less on
// paper -- made both kernels ~30 % SLOWER (fwd 6.8 -> 9.3 us, bwd 10.3 -> 13.4 on one box); the late scalar loads are L1 hits.)
// dx = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)) (+dres).  One wave per row (RW rows, one after the other, all requested up front),
// row held in registers (full thread-level parallelism hides HBM latency); the parameter gradients dgamma = sum_rows dy*xhat, dbeta =
// sum_rows dy are column reductions done by colreduce_kernel<RED_LN> (a second, bandwidth-bound pass over dy and x).
template <typename T, int NI = LN_MAXCH, int RW = 1, bool HALF = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, int M, int C) {
  constexpr int VEC = ST<T>::VEC;
  static_assert(!HALF || NI == 1, "two rows per wave: one chunk slot per lane");
  constexpr int LPR = HALF ? 32 : 64;
  const int lane = threadIdx.x & (LPR - 1);
  const int row0 = (blockIdx.x * (256 / LPR) + (threadIdx.x / LPR)) * RW;
  if (row0 >= M) return;
  const int nch = C / VEC;
  // all loads requested before the first use (see layernorm_fwd_kernel: the slot-by-slot form was SEVEN dependent round trips here -- x, dy
  // per slot, the statistics and gamma, then dres per slot)
  u32x4 rx[RW][NI], rdy[RW][NI], rdr[RW][NI];
  const T* __restrict__ drs = dres ? dres : dy;
  f32x4 gq[NI][VEC / 4];
  float mus[RW], rss[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const long rowc = row0 + r < M ? row0 + r : row0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      const long off = rowc * C + (ch < nch ? ch : 0) * VEC;
      rx[r][i] = ld16(x + off);
      rdy[r][i] = ld16(dy + off);
      rdr[r][i] = ld16(drs + off);           // (unconditional: a conditional load ends its block with a wait -- no dres: dy once more, an L1 hit)
    }
    mus[r] = mean[rowc]; rss[r] = rstd[rowc];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ch = lane + LPR * i;
    const int cc = (ch < nch ? ch : 0) * VEC;
#pragma unroll
    for (int q4 = 0; q4 < VEC / 4; ++q4) gq[i][q4] = *reinterpret_cast<const f32x4*>(gamma + cc + 4 * q4);
  }
  __builtin_amdgcn_sched_barrier(0);
  float xv[RW][NI][VEC], dv[RW][NI][VEC];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < NI; ++i) { unpack16<T>(rx[r][i], xv[r][i]); unpack16<T>(rdy[r][i], dv[r][i]); }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const float mu = mus[r], rs = rss[r];
    float xh[NI][VEC], gd[NI][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[i][e] = (xv[r][i][e] - mu) * rs;
          gd[i][e] = dv[r][i][e] * gq[i][e / 4][e % 4];
          s1 += gd[i][e];
          s2 += gd[i][e] * xh[i][e];
        }
      }
    }
    s1 = row_sum<LPR>(s1) / C;
    s2 = row_sum<LPR>(s2) / C;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = lane + LPR * i;
      if (ch < nch) {
        float o[VEC];
        if (dres) unpack16<T>(rdr[r][i], o);
        else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] += rs * (gd[i][e] - s1 - xh[i][e] * s2);
        st16(dx + (long)row * C + ch * VEC, pack16<T>(o));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Column reductions over [M, C] rows (channels fastest): sum (and optionally sum of squares, or sum of
// dy'*(1, xhat) for BN backward).  A thread owns one 16-byte channel chunk and strides over rows.
// ---------------------------------------------------------------------------------------------------
enum { RED_SUM = 0, RED_SUM_SQ = 1, RED_BN_BWD = 2, RED_PROD = 3, RED_LN = 4,
       RED_PRESUM = 5 };   // a = f32 partial rows [R][2C] written by a producer's epilogue (sum | sum of squares): only summed

// Grid (row blocks, images, column slabs).  A slab is NCHS <= 16 consecutive 16-byte channel chunks; a block covers 256 / NCHS rows of
// it at a time.  Block totals go to part[] (plain stores); the block that arrives last at its (image, slab) counter sums the row
// blocks' partials in a fixed order and writes out (deterministic: no float atomics).
// optional BatchNorm finalize folded into the last workgroup of a RED_SUM_SQ reduction (gamma == nullptr: none): batch mean / biased
// variance -> scale_shift, mean_invstd, running statistics (unbiased variance), num_batches_tracked += 1 -- what bn_finalize_kernel does
// in its own launch (and a torch kernel for the counter)
// Up to 4 BatchNorms of gc channels each may share one reduction over 4*gc channels (the e-ASPP branches): channel c belongs to
// parameter group c / gc.
struct BnFin {
  const float* gamma[4]; const float* beta[4]; float* rmean[4]; float* rvar[4]; long long* nbt[4]; float* ss; float* mi;
  float eps, momentum;
  int gc;       // channels per parameter group (== C for a single BatchNorm)
  long rows;    // RED_PRESUM: the number of samples behind the partial rows (the reduction's own row count otherwise: 0)
};
constexpr int RED_SLAB_CHUNKS = 16;
#ifndef SPG_RED_MAX_GX     // (tools/ builds may override: the last workgroup's fixed-order finish grows with the number of row blocks)
#define SPG_RED_MAX_GX 256          // (with 8 row loads in flight per thread: 256 x 32 beats 512 x 16 / 384 x 24 / 128 x 64 / 1024 x 8 --
#endif                               //  BatchNorm backward reduce + apply 55.9 -> 51.4 us, statistics 18.9 -> 16.9 us)
#ifndef SPG_RED_ROWS_PER_THREAD
#define SPG_RED_ROWS_PER_THREAD 32
#endif
constexpr int RED_MAX_GX = SPG_RED_MAX_GX;
constexpr int RED_MAX_BLOCKS = 2048;

template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const float* __restrict__ p0, const float* __restrict__ p1,
                                                        float* __restrict__ out, float* __restrict__ out1, long M, int C, int lda,
                                                        long rows_per_block, long img_rows, int relu, int nchs,
                                                        float* __restrict__ part, unsigned* __restrict__ counters, int accumulate,
                                                        BnFin fin) {
  constexpr int VEC = ST<T>::VEC;
  constexpr bool TWO = (MODE == RED_SUM_SQ || MODE == RED_BN_BWD || MODE == RED_LN || MODE == RED_PRESUM);
  constexpr int K = TWO ? 2 : 1;
  const int nch = C / VEC;
  const int rpar = 256 / nchs;  // rows handled in parallel by a block
  const int chl = threadIdx.x % nchs, rl = threadIdx.x / nchs;
  const int slab = blockIdx.z, nslabs = gridDim.z;
  const int ch = slab * nchs + chl;
  const bool active = ch < nch && rl < rpar;
  // blockIdx.y = image index for per-image reductions (img_rows > 0), else 0
  const long base = (long)blockIdx.y * img_rows;
  const long rows_total = img_rows > 0 ? img_rows : M;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows_total, r0 + rows_per_block);
  float s0[VEC], s1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  float mu[VEC], is[VEC], sc[VEC], sh[VEC];
  if constexpr (MODE == RED_BN_BWD) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = active ? ch * VEC + e : 0;
      mu[e] = p0[c]; is[e] = p0[C + c]; sc[e] = p1[c]; sh[e] = p1[C + c];
    }
  }
  if (active) {
#pragma unroll 8   // (4: 40 us for a BatchNorm-backward reduction over 8 x 192 x 192 x 64, 8: 32, 16: 44)
    for (long r = r0 + rl; r < r1; r += rpar) {
      float av[VEC];
      unpack16<T>(ld16(a + (base + r) * lda + ch * VEC), av);
      if constexpr (MODE == RED_SUM) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s0[e] += av[e];
      } else if constexpr (MODE == RED_SUM_SQ) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s0[e] += av[e]; s1[e] += av[e] * av[e]; }
      } else if constexpr (MODE == RED_PRESUM) {   // row = [sums of C channels | sums of squares of C channels], lda = 2 C
        float qv[VEC];
        unpack16<T>(ld16(a + (base + r) * lda + C + ch * VEC), qv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s0[e] += av[e]; s1[e] += qv[e]; }
      } else if constexpr (MODE == RED_LN) {  // a = dy, b = x, p0 = mean[row], p1 = rstd[row]
        float xv[VEC];
        unpack16<T>(ld16(b + (base + r) * lda + ch * VEC), xv);
        const float mu_r = p0[base + r], rs_r = p1[base + r];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s0[e] += av[e]; s1[e] += av[e] * (xv[e] - mu_r) * rs_r; }
      } else if constexpr (MODE == RED_PROD) {
        float bv[VEC];
        unpack16<T>(ld16(b + (base + r) * lda + ch * VEC), bv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s0[e] += av[e] * bv[e];
      } else {  // a = dy, b = x (pre-BN)
        float xv[VEC];
        unpack16<T>(ld16(b + (base + r) * lda + ch * VEC), xv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = (relu && (xv[e] * sc[e] + sh[e] <= 0.f)) ? 0.f : av[e];
          s0[e] += d;
          s1[e] += d * (xv[e] - mu[e]) * is[e];
        }
      }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[2][256 * 8];
  __shared__ __attribute__((aligned(16))) float fscr[256 * 4];
  __shared__ unsigned s_last;
#pragma unroll
  for (int e = 0; e < VEC; ++e) { red[0][threadIdx.x * VEC + e] = s0[e]; red[1][threadIdx.x * VEC + e] = s1[e]; }
  __syncthreads();
  const int SW = nchs * VEC;                                // columns of a slab
  const int gx = gridDim.x;
  float* mypart = part + ((((long)blockIdx.y * nslabs + slab) * gx) + blockIdx.x) * (K * SW);
  for (int c = threadIdx.x; c < SW; c += 256) {
    float t0 = 0.f, t1 = 0.f;
    for (int r = 0; r < rpar; ++r) {
      t0 += red[0][r * SW + c];
      if (TWO) t1 += red[1][r * SW + c];
    }
    st_part(mypart + c, t0);
    if (TWO) st_part(mypart + SW + c, t1);
  }
  if (!arrive_last(counters + blockIdx.y * nslabs + slab, (unsigned)gx, &s_last)) return;
  const float* pbase = part + (((long)blockIdx.y * nslabs + slab) * gx) * (K * SW);
  const int ncols = min(SW, C - slab * SW);
  float* res = &red[0][0];                                  // [K][SW] totals of this slab
  finish_rows<256>(pbase, gx, K * SW, res, fscr);
  float* o0 = out + (long)blockIdx.y * C * K + slab * SW;
  float* o1 = (out1 ? out1 : out + (long)blockIdx.y * C * K + C) + slab * SW;
  for (int cl = threadIdx.x; cl < ncols; cl += 256) {
    o0[cl] = accumulate ? o0[cl] + res[cl] : res[cl];
    if (TWO) o1[cl] = accumulate ? o1[cl] + res[SW + cl] : res[SW + cl];
  }
  if constexpr (MODE == RED_SUM_SQ || MODE == RED_PRESUM) {
    if (fin.gamma[0]) {         // BatchNorm finalize of this slab's channels, straight from the totals in LDS
      const long nrows = MODE == RED_PRESUM ? fin.rows : rows_total;
      const float Mf = (float)nrows;
      for (int cl = threadIdx.x; cl < ncols; cl += 256) {
        const int c = slab * SW + cl;
        const int grp = c / fin.gc, cg = c - grp * fin.gc;
        const float mu = res[cl] / Mf;
        const float var = fmaxf(res[SW + cl] / Mf - mu * mu, 0.f);
        if (fin.rmean[grp]) {
          fin.rmean[grp][cg] = (1.f - fin.momentum) * fin.rmean[grp][cg] + fin.momentum * mu;
          const float unb = nrows > 1 ? var * (Mf / (Mf - 1.f)) : var;
          fin.rvar[grp][cg] = (1.f - fin.momentum) * fin.rvar[grp][cg] + fin.momentum * unb;
        }
        const float is = rsqrtf(var + fin.eps);
        const float sc = fin.gamma[grp][cg] * is;
        fin.ss[c] = sc;
        fin.ss[C + c] = fin.beta[grp][cg] - mu * sc;
        if (fin.mi) { fin.mi[c] = mu; fin.mi[C + c] = is; }
      }
      if (slab == 0 && threadIdx.x < 4 && threadIdx.x * fin.gc < C && fin.nbt[threadIdx.x]) fin.nbt[threadIdx.x][0] += 1;
    }
  }
}

struct RedPlan { int nchs, nslabs, gx; long rpb; };
static inline RedPlan red_plan_rpt(long rows, int nch, int nimg, int rows_per_thread) {
  RedPlan p;
  p.nchs = nch < RED_SLAB_CHUNKS ? nch : RED_SLAB_CHUNKS;
  p.nslabs = cdiv(nch, p.nchs);
  const int rpar = 256 / p.nchs;
  const int ni = nimg > 0 ? nimg : 1;
  long want = rows / ((long)rpar * rows_per_thread);   // >= 16 rows per thread: enough blocks to stream at HBM rate, few partials
  long cap = RED_MAX_BLOCKS / ((long)ni * p.nslabs);
  if (cap > RED_MAX_GX) cap = RED_MAX_GX;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  p.rpb = (rows + want - 1) / want;
  if (p.rpb < rpar) p.rpb = rpar;
  p.gx = cdiv(rows, p.rpb);
  return p;
}
// 32 rows per thread suit the large reductions (few partials: the last workgroup's fixed-order finish stays short).  Halving them until the
// launch has 512 workgroups was measured on every user: the BatchNorm statistics / backward reductions of the head got SLOWER (58.8 -> 83.5 us
// over five launches, 202 -> 238 us over seven: the finish grows with the partials), only the per-image product of the SE backward (8 images
// x 4 slabs x 4 row blocks = 128 workgroups) gained (24.4 -> 15.0 us): `spread` is set for that mode alone.
static inline RedPlan red_plan(long rows, int nch, int nimg, int rows_per_thread = SPG_RED_ROWS_PER_THREAD, bool spread = false) {
  RedPlan p = red_plan_rpt(rows, nch, nimg, rows_per_thread);
  const int ni = nimg > 0 ? nimg : 1;
  while (spread && rows_per_thread > 4 && (long)p.gx * ni * p.nslabs < 512) {
    rows_per_thread /= 2;
    p = red_plan_rpt(rows, nch, nimg, rows_per_thread);
  }
  return p;
}
// upper bounds (any row count) of the scratch a column reduction over C channels / nimg images needs: partial floats, counters
static inline long red_ws_floats(int vec, int C, int nimg) {
  const int nch = C / vec, nchs = nch < RED_SLAB_CHUNKS ? nch : RED_SLAB_CHUNKS, nslabs = cdiv(nch, nchs), ni = nimg > 0 ? nimg : 1;
  long cap = RED_MAX_BLOCKS / ((long)ni * nslabs);
  if (cap > RED_MAX_GX) cap = RED_MAX_GX;
  if (cap < 1) cap = 1;
  return cap * ni * nslabs * 2L * nchs * vec;
}
static inline int red_counters(int vec, int C, int nimg) {
  const int nch = C / vec, nchs = nch < RED_SLAB_CHUNKS ? nch : RED_SLAB_CHUNKS;
  return cdiv(nch, nchs) * (nimg > 0 ? nimg : 1);
}

struct RedWs { float* part; long floats; unsigned* counters; };

template <typename T, int MODE>
static int launch_colreduce(const void* a, const void* b, const float* p0, const float* p1, float* out, long M, int C,
                            int lda, int nimg, long img_rows, int relu, RedWs ws, int accumulate, hipStream_t s, const char* what,
                            float* out1 = nullptr, BnFin fin = BnFin{}) {
  constexpr int VEC = ST<T>::VEC;
  if (C % VEC != 0 || lda % VEC != 0) {
    set_error("%s: C=%d (lda=%d) must be a multiple of %d", what, C, lda, VEC);
    return SPG_ERR_BAD_ARG;
  }
  const long rows = img_rows > 0 ? img_rows : M;
  // (partial rows from a producer's epilogue are few -- 1-5 k rows: a short per-thread loop and more blocks, or two blocks would walk them)
  const RedPlan p = MODE == RED_PRESUM ? red_plan(rows, C / VEC, nimg, 4) : red_plan(rows, C / VEC, nimg, SPG_RED_ROWS_PER_THREAD, MODE == RED_PROD);
  const int ni = nimg > 0 ? nimg : 1;
  constexpr bool TWO = (MODE == RED_SUM_SQ || MODE == RED_BN_BWD || MODE == RED_LN || MODE == RED_PRESUM);
  const long need = (long)p.gx * ni * p.nslabs * (TWO ? 2 : 1) * p.nchs * VEC;
  if (!ws.part || !ws.counters || ws.floats < need) {
    set_error("%s: reduction workspace of %ld floats (+ %d zeroed counters) required, got %ld", what, need, p.nslabs * ni, ws.floats);
    return SPG_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL((colreduce_kernel<T, MODE>), dim3(p.gx, ni, p.nslabs), dim3(256), 0, s, (const T*)a, (const T*)b,
                     p0, p1, out, out1, M, C, lda, p.rpb, img_rows, relu, p.nchs, ws.part, ws.counters, accumulate, fin);
  return check_launch(what);
}

// ---------------------------------------------------------------------------------------------------
// BN finalize / apply / backward-apply
// ---------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ scale_shift, float* __restrict__ mean_invstd, long M, int C,
                                   float eps, float momentum, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (training) {
    mu = stats[c] / (float)M;
    var = fmaxf(stats[C + c] / (float)M - mu * mu, 0.f);
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
  } else {
    mu = rmean[c]; var = rvar[c];
  }
  const float is = rsqrtf(var + eps);
  const float sc = gamma[c] * is;
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - mu * sc;
  if (mean_invstd) { mean_invstd[c] = mu; mean_invstd[C + c] = is; }
}

// y = act(x*scale+shift).  A thread owns one 16-byte channel chunk (its per-channel coefficients live in registers)
// and strides over rows: 1 load + 1 store per 16 bytes, 2 VALU ops per element.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                       T* __restrict__ y, long M, int C, int relu, long rows_per_block) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int rpar = 256 / nch;
  const int ch = threadIdx.x % nch, rl = threadIdx.x / nch;
  if (rl >= rpar) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = ss[ch * VEC + e]; sh[e] = ss[C + ch * VEC + e]; }
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += rpar) {
    float v[VEC];
    unpack16<T>(ld16(x + r * C + ch * VEC), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float o = v[e] * sc[e] + sh[e];
      v[e] = relu ? fmaxf(o, 0.f) : o;
    }
    st16(y + r * C + ch * VEC, pack16<T>(v));
  }
}

// dx = gamma*invstd*(dy' - s1/M - xhat*s2/M) = A*dy' + B*x + D per channel;  block 0 also does dgamma += s2, dbeta += s1
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ ss, const float* __restrict__ mi,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           T* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long M, int C, int relu,
                                                           long rows_per_block) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int rpar = 256 / nch;
  const int ch = threadIdx.x % nch, rl = threadIdx.x / nch;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      if (dbeta) atomicAdd(dbeta + c, sums[c]);
      if (dgamma) atomicAdd(dgamma + c, sums[C + c]);
    }
  }
  if (rl >= rpar) return;
  const float invM = 1.f / (float)M;
  float sc[VEC], sh[VEC], A[VEC], Bc[VEC], D[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int c = ch * VEC + e;
    sc[e] = ss[c]; sh[e] = ss[C + c];
    const float mu = mi[c], is = mi[C + c], g = gamma[c];
    const float s1 = sums[c] * invM, s2 = sums[C + c] * invM;
    A[e] = g * is;
    Bc[e] = -g * is * is * s2;
    D[e] = -g * is * s1 + g * is * is * mu * s2;
  }
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += rpar) {
    float dv[VEC], xv[VEC];
    unpack16<T>(ld16(dy + r * C + ch * VEC), dv);
    unpack16<T>(ld16(x + r * C + ch * VEC), xv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = (relu && (xv[e] * sc[e] + sh[e] <= 0.f)) ? 0.f : dv[e];
      dv[e] = A[e] * d + Bc[e] * xv[e] + D[e];
    }
    st16(dx + r * C + ch * VEC, pack16<T>(dv));
  }
}

static inline long bn_rows_per_block(long M, int nch) {
  const long rpar = 256 / nch;
  long rpb = (M + 4095) / 4096;          // up to ~4096 blocks
  if (rpb < rpar * 4) rpb = rpar * 4;    // at least 4 rows per thread
  return (rpb + rpar - 1) / rpar * rpar;
}

static inline int ew_grid(long n_items) {
  long g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace spg

using namespace spg;

#define DISPATCH_T(dtype, CALL_BF, CALL_F32) ((dtype) == SPG_BF16 ? (CALL_BF) : (CALL_F32))

extern "C" int spg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int M, int C, float eps, spg_stream_t stream) {
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(M > 0 && C > 0 && C % vec == 0 && C / vec <= 64 * LN_MAXCH, "layernorm_fwd: C=%d must be a multiple of %d and <= %d", C, vec, 64 * LN_MAXCH * vec);
  hipStream_t s = (hipStream_t)stream;
  const int ni = cdiv(C / vec, 64);
#define SPG_LNF(T_, NI_) hipLaunchKernelGGL((layernorm_fwd_kernel<T_, NI_>), dim3(cdiv(M, 4)), dim3(256), 0, s, (const T_*)x, gamma, beta, (T_*)y, mean, rstd, M, C, eps)
#define SPG_LNF_RW(T_, RW_) hipLaunchKernelGGL((layernorm_fwd_kernel<T_, 1, RW_>), dim3(cdiv(M, 4 * RW_)), dim3(256), 0, s, (const T_*)x, gamma, beta, (T_*)y, mean, rstd, M, C, eps)
#ifndef SPG_LN_RW_ROWS
#define SPG_LN_RW_ROWS 32768     // rows from which a wave takes four rows of a one-slot (C <= 512 bf16) LayerNorm (tools/ A/B builds: a huge value = never)
#endif
  if (dtype == SPG_BF16 && ni <= 1 && M >= SPG_LN_RW_ROWS) {      // many short rows (stage 1: 73728 x 144: 24.5 -> 22.2 us; 18432 x 288 gains nothing)
#ifndef SPG_LN_HALF      // (tools/ A/B builds: 0 = a row per wave whatever its length)
#define SPG_LN_HALF 1
#endif
    if (SPG_LN_HALF && C / vec <= 32)     // ... and rows of at most 32 chunks two to a wave
      hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, 1, 4, true>), dim3(cdiv(M, 8 * 4)), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps);
    else
      SPG_LNF_RW(bf16_t, 4);
    return check_launch("layernorm_fwd");
  }
  if (dtype == SPG_BF16) { if (ni <= 1) SPG_LNF(bf16_t, 1); else if (ni == 2) SPG_LNF(bf16_t, 2); else if (ni == 3) SPG_LNF(bf16_t, 3); else SPG_LNF(bf16_t, LN_MAXCH); }
  else { if (ni <= 1) SPG_LNF(float, 1); else if (ni == 2) SPG_LNF(float, 2); else if (ni == 3) SPG_LNF(float, 3); else SPG_LNF(float, LN_MAXCH); }
#undef SPG_LNF
#undef SPG_LNF_RW
  return check_launch("layernorm_fwd");
}

extern "C" int spg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int M, int C,
                                 float* red_ws, long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(M > 0 && C > 0 && C % vec == 0 && C / vec <= 64 * LN_MAXCH, "layernorm_bwd: bad C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  const int ni = cdiv(C / vec, 64);
#define SPG_LNB(T_, NI_) hipLaunchKernelGGL((layernorm_bwd_kernel<T_, NI_>), dim3(cdiv(M, 4)), dim3(256), 0, s, (const T_*)dy, (const T_*)x, gamma, mean, rstd, (const T_*)dres, (T_*)dx, M, C)
  // (several rows per wave, as the forward does for many short rows: 73728 x 144 backward 35.3 us with four rows per wave, 26.2 with two,
  // 26.4 with one; 18432 x 288: 11.4 / 11.6 -- nothing to gain, the one-row kernel stays)
  if (dtype == SPG_BF16 && SPG_LN_HALF && C / vec <= 32 && M >= 8)      // short rows (stage 1): two to a wave
    hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 1, 1, true>), dim3(cdiv(M, 8)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, C);
  else if (dtype == SPG_BF16) { if (ni <= 1) SPG_LNB(bf16_t, 1); else if (ni == 2) SPG_LNB(bf16_t, 2); else if (ni == 3) SPG_LNB(bf16_t, 3); else SPG_LNB(bf16_t, LN_MAXCH); }
  else { if (ni <= 1) SPG_LNB(float, 1); else if (ni == 2) SPG_LNB(float, 2); else if (ni == 3) SPG_LNB(float, 3); else SPG_LNB(float, LN_MAXCH); }
#undef SPG_LNB
  int rc = check_launch("layernorm_bwd");
  if (rc || (!dgamma && !dbeta)) return rc;
  SPG_REQUIRE(dgamma && dbeta, "layernorm_bwd: dgamma and dbeta must both be given");
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_LN>(dy, x, mean, rstd, dbeta, M, C, C, 0, 0, 0, ws, 1, s, "layernorm_bwd(params)", dgamma)),
                    (launch_colreduce<float, RED_LN>(dy, x, mean, rstd, dbeta, M, C, C, 0, 0, 0, ws, 1, s, "layernorm_bwd(params)", dgamma)));
}

// ---------------------------------------------------------------------------------------------------
// Batched LayerNorm parameter gradients: dgamma[j] += sum_rows dy * xhat, dbeta[j] += sum_rows dy for up to 48 LayerNorms in ONE launch.
// They only feed the optimizer, so the trunk backward collects them and issues two launches per step instead of 96 (every kernel
// costs ~4.5 us of launch floor on this GPU).  Jobs travel in the kernel-argument struct: no device table.
// ---------------------------------------------------------------------------------------------------
constexpr int LN_BATCH_MAX = 48;
#ifndef SPG_LN_BATCH_BLOCKS
#define SPG_LN_BATCH_BLOCKS 32   // (16: 178 us per batched launch, 24: 158, 32: 150, 64: 157, 128: 198)
#endif
constexpr int LN_BATCH_BLOCKS = SPG_LN_BATCH_BLOCKS;   // row blocks per job (upper bound): the partials of a job are <= LN_BATCH_BLOCKS x 2C floats
struct LnJob {
  const void* dy; const void* x; const float* mean; const float* rstd; float* dgamma; float* dbeta;
  int M, C, ld, block0, rpb, nblk, poff;  // C: columns of this job (<= 256 16-byte chunks), ld: row stride; block0: first block; rpb: rows per block;
};                                        // nblk: its blocks; poff: float offset of its partials
struct LnBatch { LnJob job[LN_BATCH_MAX]; int njobs; };

template <typename T>
__global__ __launch_bounds__(256) void ln_param_batch_kernel(LnBatch bt, float* __restrict__ part, unsigned* __restrict__ counters) {
  constexpr int VEC = ST<T>::VEC;
  int j = 0;
#pragma unroll 1
  for (int i = 1; i < bt.njobs; ++i) if ((int)blockIdx.x >= bt.job[i].block0) j = i;
  const LnJob& jb = bt.job[j];
  const T* dy = reinterpret_cast<const T*>(jb.dy);
  const T* x = reinterpret_cast<const T*>(jb.x);
  const int C = jb.C, nch = C / VEC;
  const int rpar = 256 / nch;
  const int ch = threadIdx.x % nch, rl = threadIdx.x / nch;
  const int bl = (int)blockIdx.x - jb.block0;
  const long r0 = (long)bl * jb.rpb;
  const long r1 = min((long)jb.M, r0 + jb.rpb);
  float s0[VEC], s1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  if (rl < rpar) {
    for (long r = r0 + rl; r < r1; r += rpar) {
      float dv[VEC], xv[VEC];
      unpack16<T>(ld16(dy + r * jb.ld + ch * VEC), dv);
      unpack16<T>(ld16(x + r * jb.ld + ch * VEC), xv);
      const float mu = jb.mean[r], rs = jb.rstd[r];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s0[e] += dv[e]; s1[e] += dv[e] * (xv[e] - mu) * rs; }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[2][256 * 8];
  __shared__ __attribute__((aligned(16))) float fscr[256 * 4];
  __shared__ unsigned s_last;
#pragma unroll
  for (int e = 0; e < VEC; ++e) { red[0][threadIdx.x * VEC + e] = s0[e]; red[1][threadIdx.x * VEC + e] = s1[e]; }
  __syncthreads();
  float* mypart = part + jb.poff + (long)bl * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t0 = 0.f, t1 = 0.f;
    for (int r = 0; r < rpar; ++r) { t0 += red[0][r * nch * VEC + c]; t1 += red[1][r * nch * VEC + c]; }
    st_part(mypart + c, t0);
    st_part(mypart + C + c, t1);
  }
  if (!arrive_last(counters + j, (unsigned)jb.nblk, &s_last)) return;
  float* res = &red[0][0];                                  // [2][C] totals (C <= 2048: exactly red's size)
  finish_rows<256>(part + jb.poff, jb.nblk, 2 * C, res, fscr);
  for (int c = threadIdx.x; c < C; c += 256) { jb.dbeta[c] += res[c]; jb.dgamma[c] += res[C + c]; }
}

/* scratch: LN_BATCH_BLOCKS * 2 * sum(C[i]) floats of partials and njobs zeroed counters */
extern "C" long spg_layernorm_param_grads_batch_workspace_floats(int njobs, const int* C) {
  long t = 0;
  for (int i = 0; i < njobs; ++i) t += (long)LN_BATCH_BLOCKS * 2 * C[i];
  return t;
}

extern "C" int spg_layernorm_param_grads_batch(int dtype, int njobs, const void* const* dy, const void* const* x, const float* const* mean,
                                               const float* const* rstd, float* const* dgamma, float* const* dbeta, const int* M,
                                               const int* C, const int* ld, float* red_ws, long red_ws_floats, unsigned* red_counters_,
                                               spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_F32 || dtype == SPG_BF16, "layernorm_param_grads_batch: bad dtype %d", dtype);
  SPG_REQUIRE(njobs >= 1 && njobs <= LN_BATCH_MAX, "layernorm_param_grads_batch: 1..%d jobs, got %d", LN_BATCH_MAX, njobs);
  SPG_REQUIRE(red_ws && red_counters_, "layernorm_param_grads_batch: reduction workspace and counters required");
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  LnBatch bt;
  int blocks = 0;
  long poff = 0;
  for (int i = 0; i < njobs; ++i) {
    SPG_REQUIRE(M[i] > 0 && C[i] > 0 && C[i] % vec == 0 && C[i] / vec <= 256 && ld[i] >= C[i] && ld[i] % vec == 0,
                "layernorm_param_grads_batch: job %d: bad M=%d C=%d ld=%d (C <= %d columns per job: split wider rows)", i, M[i], C[i], ld[i], 256 * vec);
    LnJob& jb = bt.job[i];
    jb.dy = dy[i]; jb.x = x[i]; jb.mean = mean[i]; jb.rstd = rstd[i]; jb.dgamma = dgamma[i]; jb.dbeta = dbeta[i];
    jb.M = M[i]; jb.C = C[i]; jb.ld = ld[i];
    const int rpar = 256 / (C[i] / vec);
    int rpb = cdiv(M[i], LN_BATCH_BLOCKS);           // ~64 blocks per job: enough rows in flight, few partials per column
    if (rpb < rpar * 8) rpb = rpar * 8;
    jb.rpb = rpb; jb.block0 = blocks;
    jb.nblk = cdiv(M[i], rpb);
    jb.poff = (int)poff;
    poff += (long)jb.nblk * 2 * C[i];
    blocks += jb.nblk;
  }
  SPG_REQUIRE(poff <= red_ws_floats, "layernorm_param_grads_batch: workspace of %ld floats required, got %ld", poff, red_ws_floats);
  for (int i = njobs; i < LN_BATCH_MAX; ++i) bt.job[i] = bt.job[njobs - 1];
  bt.njobs = njobs;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(ln_param_batch_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt, red_ws, red_counters_);
  else hipLaunchKernelGGL(ln_param_batch_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt, red_ws, red_counters_);
  return check_launch("layernorm_param_grads_batch");
}

/* scratch a column reduction over C channels (per image when nimg > 0) needs, for any row count: floats of partials, and counters
 * (32-bit words that must be ZERO before their first use; every launch leaves them zero again) */
extern "C" long spg_reduce_workspace_floats(int dtype, int C, int nimg) { return red_ws_floats(dtype == SPG_BF16 ? 8 : 4, C, nimg); }
extern "C" int spg_reduce_counters(int dtype, int C, int nimg) { return red_counters(dtype == SPG_BF16 ? 8 : 4, C, nimg); }

extern "C" int spg_colsum(int dtype, const void* x, float* out, int M, int C, int ldx, int accumulate, float* red_ws, long red_ws_floats,
                          unsigned* red_counters_, spg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_SUM>(x, nullptr, nullptr, nullptr, out, M, C, ldx, 0, 0, 0, ws, accumulate, s, "colsum")),
                    (launch_colreduce<float, RED_SUM>(x, nullptr, nullptr, nullptr, out, M, C, ldx, 0, 0, 0, ws, accumulate, s, "colsum")));
}

extern "C" int spg_bn_stats(int dtype, const void* x, float* stats, long M, int C, float* red_ws, long red_ws_floats,
                            unsigned* red_counters_, spg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats")),
                    (launch_colreduce<float, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats")));
}

// batch statistics AND the finalize in one launch (training mode): stats f32 [2C] scratch (overwritten), the rest as bn_finalize;
// num_batches_tracked (int64, may be NULL) is incremented
extern "C" int spg_bn_stats_finalize(int dtype, const void* x, float* stats, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, long long* num_batches_tracked, float* scale_shift, float* mean_invstd, long M,
                                     int C, float eps, float momentum, float* red_ws, long red_ws_floats, unsigned* red_counters_,
                                     spg_stream_t stream) {
  SPG_REQUIRE(gamma && beta && scale_shift && stats, "bn_stats_finalize: gamma, beta, stats and scale_shift are required");
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  BnFin fin{};
  fin.gamma[0] = gamma; fin.beta[0] = beta; fin.rmean[0] = running_mean; fin.rvar[0] = running_var; fin.nbt[0] = num_batches_tracked;
  fin.ss = scale_shift; fin.mi = mean_invstd; fin.eps = eps; fin.momentum = momentum; fin.gc = C;
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats_finalize", nullptr, fin)),
                    (launch_colreduce<float, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats_finalize", nullptr, fin)));
}

// the same from PARTIAL statistics: part f32 [R][2C] rows of (sums | sums of squares) over disjoint sample sets (the epilogue of
// spg_conv3x3_fwd_stats writes them), M samples in all.  One launch: fixed-order sum of the rows + finalize; the conv output is not read.
extern "C" int spg_bn_stats_finalize_part(const float* part, long R, float* stats, const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, long long* num_batches_tracked, float* scale_shift, float* mean_invstd, long M,
                                          int C, float eps, float momentum, float* red_ws, long red_ws_floats, unsigned* red_counters_,
                                          spg_stream_t stream) {
  SPG_REQUIRE(part && gamma && beta && scale_shift && stats, "bn_stats_finalize_part: part, gamma, beta, stats and scale_shift are required");
  SPG_REQUIRE(R > 0 && M > 0 && C % 4 == 0, "bn_stats_finalize_part: R=%ld M=%ld C=%d", R, M, C);
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  BnFin fin{};
  fin.gamma[0] = gamma; fin.beta[0] = beta; fin.rmean[0] = running_mean; fin.rvar[0] = running_var; fin.nbt[0] = num_batches_tracked;
  fin.ss = scale_shift; fin.mi = mean_invstd; fin.eps = eps; fin.momentum = momentum; fin.gc = C; fin.rows = M;
  return launch_colreduce<float, RED_PRESUM>(part, nullptr, nullptr, nullptr, stats, R, C, 2 * C, 0, 0, 0, ws, 0, (hipStream_t)stream,
                                             "bn_stats_finalize_part", nullptr, fin);
}

// the same for four BatchNorms of C/4 channels each over one [M, C] tensor (the e-ASPP branches stored branch-major): HOST arrays of 4 pointers
extern "C" int spg_bn_stats_finalize4(int dtype, const void* x, float* stats, const float* const* gamma4, const float* const* beta4,
                                      float* const* running_mean4, float* const* running_var4, long long* const* num_batches_tracked4,
                                      float* scale_shift, float* mean_invstd, long M, int C, float eps, float momentum, float* red_ws,
                                      long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  SPG_REQUIRE(gamma4 && beta4 && scale_shift && stats && C % 4 == 0, "bn_stats_finalize4: C=%d must be 4 groups", C);
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  BnFin fin{};
  for (int i = 0; i < 4; ++i) {
    SPG_REQUIRE(gamma4[i] && beta4[i], "bn_stats_finalize4: group %d has no parameters", i);
    fin.gamma[i] = gamma4[i]; fin.beta[i] = beta4[i];
    fin.rmean[i] = running_mean4 ? running_mean4[i] : nullptr; fin.rvar[i] = running_var4 ? running_var4[i] : nullptr;
    fin.nbt[i] = num_batches_tracked4 ? num_batches_tracked4[i] : nullptr;
  }
  fin.ss = scale_shift; fin.mi = mean_invstd; fin.eps = eps; fin.momentum = momentum; fin.gc = C / 4;
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats_finalize4", nullptr, fin)),
                    (launch_colreduce<float, RED_SUM_SQ>(x, nullptr, nullptr, nullptr, stats, M, C, C, 0, 0, 0, ws, 0, s, "bn_stats_finalize4", nullptr, fin)));
}

extern "C" int spg_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* scale_shift, float* mean_invstd, long M, int C, float eps,
                               float momentum, int training, spg_stream_t stream) {
  SPG_REQUIRE(C > 0 && M > 0, "bn_finalize: empty");
  SPG_REQUIRE(training || (running_mean && running_var), "bn_finalize: eval mode needs running stats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, stats, gamma, beta,
                     running_mean, running_var, scale_shift, mean_invstd, M, C, eps, momentum, training);
  return check_launch("bn_finalize");
}

extern "C" int spg_bn_apply(int dtype, const void* x, const float* scale_shift, void* y, long M, int C, int relu,
                            spg_stream_t stream) {
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(C % vec == 0, "bn_apply: C=%d must be a multiple of %d", C, vec);
  SPG_REQUIRE(C / vec <= 256, "bn_apply: C=%d too wide", C);
  hipStream_t s = (hipStream_t)stream;
  const long rpb = bn_rows_per_block(M, C / vec);
  const int grid = cdiv(M, rpb);
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, scale_shift, (bf16_t*)y, M, C, relu, rpb);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, scale_shift, (float*)y, M, C, relu, rpb);
  return check_launch("bn_apply");
}

extern "C" int spg_bn_bwd_reduce(int dtype, const void* dy, const void* x, const float* scale_shift,
                                 const float* mean_invstd, float* sums, long M, int C, int relu, float* red_ws, long red_ws_floats,
                                 unsigned* red_counters_, spg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_BN_BWD>(dy, x, mean_invstd, scale_shift, sums, M, C, C, 0, 0, relu, ws, 0, s, "bn_bwd_reduce")),
                    (launch_colreduce<float, RED_BN_BWD>(dy, x, mean_invstd, scale_shift, sums, M, C, C, 0, 0, relu, ws, 0, s, "bn_bwd_reduce")));
}

extern "C" int spg_bn_bwd_apply(int dtype, const void* dy, const void* x, const float* scale_shift,
                                const float* mean_invstd, const float* gamma, const float* sums, void* dx, float* dgamma,
                                float* dbeta, long M, int C, int relu, spg_stream_t stream) {
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(C % vec == 0, "bn_bwd_apply: C=%d must be a multiple of %d", C, vec);
  SPG_REQUIRE(C / vec <= 256, "bn_bwd_apply: C=%d too wide", C);
  hipStream_t s = (hipStream_t)stream;
  const long rpb = bn_rows_per_block(M, C / vec);
  const int grid = cdiv(M, rpb);
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, scale_shift, mean_invstd, gamma, sums, (bf16_t*)dx, dgamma, dbeta, M, C, relu, rpb);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)x, scale_shift, mean_invstd, gamma, sums, (float*)dx, dgamma, dbeta, M, C, relu, rpb);
  return check_launch("bn_bwd_apply");
}

// per-image column mean numerators: out[b][c] = sum_hw x[b][hw][c]   (caller divides)
extern "C" int spg_gap_sum(int dtype, const void* x, float* out, int B, long HW, int C, float* red_ws, long red_ws_floats,
                           unsigned* red_counters_, spg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_SUM>(x, nullptr, nullptr, nullptr, out, B * HW, C, C, B, HW, 0, ws, 0, s, "gap_sum")),
                    (launch_colreduce<float, RED_SUM>(x, nullptr, nullptr, nullptr, out, B * HW, C, C, B, HW, 0, ws, 0, s, "gap_sum")));
}
// per-image channel products: out[b][c] = sum_hw a[b][hw][c]*b[b][hw][c]   (SE scale gradient)
extern "C" int spg_chan_prod_sum(int dtype, const void* a, const void* b, float* out, int B, long HW, int C, float* red_ws,
                                 long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const RedWs ws{red_ws, red_ws_floats, red_counters_};
  return DISPATCH_T(dtype, (launch_colreduce<bf16_t, RED_PROD>(a, b, nullptr, nullptr, out, B * HW, C, C, B, HW, 0, ws, 0, s, "chan_prod_sum")),
                    (launch_colreduce<float, RED_PROD>(a, b, nullptr, nullptr, out, B * HW, C, C, B, HW, 0, ws, 0, s, "chan_prod_sum")));
}
