// MFMA GEMM family for gfx950.
//
//   gemm_nt : C[M,N] = epi(X[M,K] . W[N,K]^T)   forward + dgrad of every Linear / 1x1 conv, and (conv3x3
//             gather on X) forward + dgrad of every 3x3 convolution as an implicit GEMM over NHWC.
//   gemm_tn : dW[N,K] += dY[M,N]^T . X[M,K]     weight gradients (reduction over the M rows), operands are
//             transposed through registers on their way into LDS; split over M with f32 atomics.
//
// Tile 128(m) x 128(n), 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 blocks, f32 accumulators.
// LDS rows are 128 bytes (64 bf16 / 32 f32 of K), 16-byte chunks XOR-swizzled so the fragment reads
// (ds_read_b128 for bf16, ds_read_b32 for f32) are bank-conflict free / 2-way.  Register-staged double
// buffering: global loads for tile t+1 are issued before the MFMAs of tile t, written to the other LDS
// buffer after them, one barrier per K-tile.
//
// MFMA operand roles: "A" = weight rows (n), "B" = activation rows (m)  =>  D[n][m]: a lane holds 4
// consecutive n for one m, so the epilogue reads bias / writes C as 8- or 16-byte vectors.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

// SPG_DEV_KERNELS (tools/ builds only, `python spegnet_amd/build.py --dev`): the superseded / experimental kernel families kept for A/B
// measurements (register-staged NT kernel, 4-wave DMA kernel, deferred-epilogue and ablation instances of the pipelined kernels, the
// 8-wave staged wgrad kernel, the 256 x 128 "wide" grouped wgrad kernel) and the SPG_* environment switches that select them
// (DESIGN.md 3.1).  The product library is built WITHOUT it: it contains none of those kernels and reads no environment variable.
namespace spg {

#ifdef SPG_DEV_KERNELS
static int dev_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#endif

constexpr int BM = 128, BN = 128, ROWB = 128;  // tile rows, LDS row bytes
constexpr int NT_THREADS = 256;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int SUB = 2;  // MFMA k-steps per LDS row (2 x 32)
  using Frag = bf16x8_t;
  __device__ static __forceinline__ Frag load(const char* row, int sw, int s, int q) {
    return *reinterpret_cast<const Frag*>(row + (((4 * s + q) ^ sw) << 4));
  }
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int SUB = 8;  // 8 x 4
  using Frag = float;
  __device__ static __forceinline__ Frag load(const char* row, int sw, int s, int q) {
    return *reinterpret_cast<const float*>(row + ((s ^ sw) << 4) + (q << 2));
  }
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// XCD-aware bijective remap: blocks sharing blockIdx%8 (one XCD) get a contiguous range of tile ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

struct ConvGeom {
  int B, H, W, Ci;  // NHWC input of a 3x3/pad1/stride1 convolution
};
// conv_halo.hip: bf16 3x3 convolution with an LDS-resident input tile; returns 1 when the problem is outside its domain
int launch_conv3x3_halo(const void* X, const void* Wp, void* C, const float* bias, int B, int H, int W, int Ci, int Co, int ldc,
                        int cus, int force_bn, hipStream_t s, float* stats);
// tn_block.hip: the weight gradients of several trunk blocks as whole 256 x 192 blocks of dW, one per workgroup, added straight into dW
// (every N, K a multiple of 192, at most `cus` blocks); 1 = outside the domain.  tn_blocks_count: blocks a problem set makes (-1: outside)
int launch_tn_blocks_direct(int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, int M, const int* N,
                            const int* K, const int* ldy, const int* ldx, const int* ldw, int cus, hipStream_t s, int overwrite, float* sq_part);
long tn_blocks_count(int njobs, int M, const int* N, const int* K);
#ifdef SPG_DEV_KERNELS
// nt_wide.hip (dev builds): dense bf16 NT GEMM on 192-column tiles of variable height (act = PIPE_ACT_* code); returns 1 when the problem is outside its domain
int launch_nt_wide(const void* X, const void* W, void* C, const float* bias, const void* Hh, void* C2, int act, int M, int N, int K, int ldx,
                   int ldc, int cus, int dbg, hipStream_t s);
// tn_block.hip (dev builds): grouped weight gradients with 256 x 192 blocks per workgroup (every N, K a multiple of 192); returns 1 when not applicable
int launch_tn_block_group(int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, int M, const int* N,
                          const int* K, const int* ldy, const int* ldx, const int* ldw, void* workspace, long workspace_bytes, int cus,
                          hipStream_t s);
long tn_block_workspace_bytes(int cus);
#endif

// Branch-free operand loads: raw buffer loads with hardware bounds checking (out-of-range offsets return 0), so
// hipcc keeps every load of a K-tile in flight behind a counted vmcnt instead of branching around each one
// (cdna_hip_programming.md §5 "Three .s-level traps" (c)).
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned bufvec_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  bufvec_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return u32x4{v[0], v[1], v[2], v[3]};
}
constexpr unsigned OOB = 0xFFFFFFF0u;

// LDS-DMA through inline asm: the compiler then knows of no LDS writes in flight and does not guard later LDS reads with
// s_waitcnt vmcnt(0) (it does exactly that before ds_read_b64_tr_b16 when the DMA is issued through the builtin, draining the whole
// fill pipeline every step).  Ordering is the kernel's job: counted vmcnt waits + barriers, as documented at each call site.
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_words_t;
__device__ __forceinline__ rsrc_words_t make_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  rsrc_words_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16_asm(rsrc_words_t rsrc, unsigned lds_addr, unsigned voff) {   // 64 lanes x 16 B -> LDS[lds_addr ..+1024)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// byte offset of the 16-byte chunk (row m, K offset k0) of the (possibly gathered) X operand; OOB when outside.
template <typename T, bool CONV>
__device__ __forceinline__ unsigned x_chunk_off(int m, int k0, int ldx, const ConvGeom& g, int py, int px) {
  if constexpr (!CONV) {
    return (unsigned)(((long)m * ldx + k0) * (long)sizeof(T));
  } else {
    const int tap = k0 / g.Ci, ci = k0 - tap * g.Ci;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const bool in = (unsigned)(py + dy) < (unsigned)g.H && (unsigned)(px + dx) < (unsigned)g.W;
    const unsigned off = (unsigned)((((long)m + (long)dy * g.W + dx) * g.Ci + ci) * (long)sizeof(T));
    return in ? off : OOB;
  }
}

// wave tile = 64 (n) x 16*MI (m); the wave's first m row inside the block tile is mrow0
template <typename T, int MI = 4, int NB = 4, bool PREFETCH = false>
__device__ __forceinline__ void mma_tile(const char* __restrict__ Ws, const char* __restrict__ Xs, int wn, int wm,
                                         int lane, f32x4 (&acc)[NB][MI]) {
  using M_ = Mma<T>;
  const int r = lane & 15, q = lane >> 4;
  if constexpr (sizeof(T) == 2 && PREFETCH) {
    // every fragment of the step is requested before the first MFMA: one exposed LDS latency per step instead of one per
    // operand group (the reads return in order, so the MFMAs start as soon as the first k-half has arrived)
    typename M_::Frag a[2][NB], b[2][MI];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rowb = wm * (16 * MI) + i * 16 + r;
        b[s][i] = M_::load(Xs + rowb * ROWB, rowb & 7, s, q);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int rowa = wn * (16 * NB) + i * 16 + r;
        a[s][i] = M_::load(Ws + rowa * ROWB, rowa & 7, s, q);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ni = 0; ni < NB; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = M_::mma(a[s][ni], b[s][mi], acc[ni][mi]);
    return;
  }
#pragma unroll
  for (int s = 0; s < M_::SUB; ++s) {
    typename M_::Frag a[NB], b[MI];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int rowa = wn * (16 * NB) + i * 16 + r;
      a[i] = M_::load(Ws + rowa * ROWB, rowa & 7, s, q);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int rowb = wm * (16 * MI) + i * 16 + r;
      b[i] = M_::load(Xs + rowb * ROWB, rowb & 7, s, q);
    }
#pragma unroll
    for (int ni = 0; ni < NB; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = M_::mma(a[ni], b[mi], acc[ni][mi]);
  }
}

struct NtEpi {
  const float* bias;
  const void* residual;
  const void* gelu_h;
  void* C2;
  int act;
};

#ifdef SPG_DEV_KERNELS
#include "dev/gemm_nt_regstage.inc"
#endif  // SPG_DEV_KERNELS (register-staged NT kernel)


// ------------------------------------------------------------------------------------------------
// gemm_nt, LDS-DMA pipeline (the default): persistent workgroups (one per CU) walk their tiles; operands stream
// global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, so hipcc has no register dependences to serialise)
// into a 3-stage ring; a K-step is {counted vmcnt -> raw s_barrier -> issue DMA for step+2 -> 32 MFMAs per wave}.
// The ring runs across tile boundaries, so the next tile's first K-steps land while the current tile's epilogue runs.
// LDS image per stage is identical to the register-staged kernel: 128-byte rows, 16-byte chunk p of row r holds logical
// chunk p ^ (r & 7); the DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE offset
// (cdna_hip_programming.md rule 21).  Out-of-range rows / the K tail use out-of-range buffer offsets (hardware zero fill).
// ------------------------------------------------------------------------------------------------
constexpr int DMA_STAGES = 4;
constexpr int DMA_STAGE_BYTES = 2 * 128 * ROWB;      // X tile + W tile
constexpr int DMA_SLAB_BYTES = 4 * 16 * 68 * 4;      // per-wave epilogue slabs (16 rows x 64 f32 + pad)
constexpr int DMA_LDS_BYTES = DMA_STAGES * DMA_STAGE_BYTES + DMA_SLAB_BYTES;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// DBG: 0 = product kernel; 1 = skip the DMA issue (times MFMA + LDS reads alone); 2 = skip the MFMAs (times the fill
// pipeline alone).  1 and 2 produce wrong results by construction and are reachable only through SPG_GEMM_DEBUG.
// NB = 16-row n blocks per wave: block tile width BN_ = 32*NB (128 / 96 / 64), so narrow outputs (N = 64 convolutions, N = 576 trunk
// projections) do not multiply zero rows or strand CUs on too few tiles.
template <typename T, bool CONV, int DBG = 0, int WM = 2, int NB = 4>
__global__ __launch_bounds__(128 * WM) void gemm_nt_dma_kernel(const T* __restrict__ X, const T* __restrict__ W,
                                                                 T* __restrict__ C, NtEpi epi, int M, int N, int K, int ldx,
                                                                 int ldc, ConvGeom g, int tiles_n, int ntiles,
                                                                 unsigned xbytes, unsigned wbytes) {
  constexpr int VEC = ST<T>::VEC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int MI = 8 / WM;            // 16-row m blocks per wave (wave tile 64 n x 16*MI m)
  constexpr int NW = 2 * WM;            // waves per workgroup
  constexpr int NP = 16 / NW;           // DMA pieces (1 KiB) per operand per wave per stage
  constexpr int STAGES = WM == 2 ? DMA_STAGES : 3;
  constexpr int BN_ = 32 * NB;          // tile width in n
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int nkt = (K + BK - 1) / BK;
  const int G = (int)gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + G - 1) / G;
  const int total = my_tiles * nkt;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(X, xbytes), wr = make_rsrc(W, wbytes);

  // ---- DMA issue stream state (runs DMA_STAGES-1 steps ahead of the MFMAs)
  const int lrow = lane >> 3;             // row within an 8-row DMA piece
  const int lp = lane & 7;                // physical 16-byte chunk
  int is_j = -1, is_m0 = 0, is_n0 = 0;
  int py[NP], px[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) { py[i] = 0; px[i] = 0; }
  auto issue = [&](int gs) {
    const int j = gs / nkt, kt = gs - j * nkt;
    if (j != is_j) {
      is_j = j;
      const int tile = first + j * G;
      const int tn = tile % tiles_n, tm = tile / tiles_n;
      is_m0 = tm * BM; is_n0 = tn * BN_;
      if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int m = is_m0 + (8 * NW) * i + wave * 8 + lrow;
          const int hw = g.H * g.W;
          const int b = m / hw, rem = m - b * hw;
          py[i] = rem / g.W; px[i] = rem - py[i] * g.W;
        }
      }
    }
    char* st = smem + (gs % STAGES) * DMA_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int prow = (8 * NW) * i + wave * 8;     // first row of this wave-instruction's 1 KiB piece
      const int row = prow + lrow;
      const int k0 = kt * BK + ((lp ^ (row & 7)) * VEC);
      const bool kin = k0 < K;
      const unsigned xo = x_chunk_off<T, CONV>(is_m0 + row, k0, ldx, g, py[i], px[i]);
      const unsigned wo = (unsigned)(((long)(is_n0 + row) * K + k0) * (long)sizeof(T));
      const bool win = kin && row < BN_;            // rows >= BN_ of the 128-row W area are never read: zero fill, no traffic
      if constexpr (DBG != 1 && DBG != 3 && DBG != 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + prow * ROWB), 16, kin ? xo : OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + BM * ROWB + prow * ROWB), 16, win ? wo : OOB, 0, 0, 0);
      } else {
        asm volatile("" :: "v"(xo), "v"(wo));
      }
    }
  };

  f32x4 acc[NB][MI];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int gi = 0;
  for (; gi < STAGES - 1 && gi < total; ++gi) issue(gi);

  constexpr int EPS = 68;
  float* slab = reinterpret_cast<float*>(smem + STAGES * DMA_STAGE_BYTES) + wave * (16 * EPS);
  const int r15 = lane & 15, q = lane >> 4;
  const T* R = reinterpret_cast<const T*>(epi.residual);
  const T* Hh = reinterpret_cast<const T*>(epi.gelu_h);
  T* C2 = reinterpret_cast<T*>(epi.C2);
  const bool vec_ok = (ldc % 8 == 0);

  int kt = 0, j = 0;
  for (int gc = 0; gc < total; ++gc) {
    // step gc's 8 DMA pieces (per wave) must have landed; newer groups may stay in flight
    {
      const int ahead = gi - gc - 1;   // DMA groups younger than step gc's (2*NP pieces each per wave)
      if constexpr (NP == 4) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (gi < total) { issue(gi); ++gi; }
    const char* st = smem + (gc % STAGES) * DMA_STAGE_BYTES;
    if constexpr (DBG != 2) mma_tile<T, MI, NB, (WM == 4)>(st + BM * ROWB, st, wn, wm, lane, acc);
    if (++kt == nkt) {
      if constexpr (DBG == 3) {  // ablation: no fill, no epilogue (the never-true store keeps the MFMAs live)
#pragma unroll
        for (int quarter = 0; quarter < MI; ++quarter)
#pragma unroll
          for (int ni = 0; ni < NB; ++ni) {
            if (M < 0) *reinterpret_cast<f32x4*>(slab + r15 * EPS + ni * 16 + q * 4) = acc[ni][quarter];
            acc[ni][quarter] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        kt = 0; ++j;
        continue;
      }
      // ---- epilogue of tile j (per-wave slab, no block barrier: waves run ahead into the next tile independently)
      const int tile = first + j * G;
      const int tn = tile % tiles_n, tm = tile / tiles_n;
      const int m0 = tm * BM, n0 = tn * BN_;
#pragma unroll
      for (int quarter = 0; quarter < MI; ++quarter) {
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
          *reinterpret_cast<f32x4*>(slab + r15 * EPS + ni * 16 + q * 4) = acc[ni][quarter];
          acc[ni][quarter] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int row = (lane >> 3) + 8 * jj, ch = lane & 7;
          const int m = m0 + wm * (16 * MI) + quarter * 16 + row;
          const int n = n0 + wn * (16 * NB) + ch * 8;
          if (m < M && n < N && ch * 8 < 16 * NB) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + row * EPS + ch * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + row * EPS + ch * 8 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const long o = (long)m * ldc + n;
            if (vec_ok && n + 7 < N) {
              if (epi.bias) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(epi.bias + n), b1 = *reinterpret_cast<const f32x4*>(epi.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
              }
              if (C2) {
                if constexpr (sizeof(T) == 2) st16(C2 + o, pack16<T>(v));
                else { st16(C2 + o, pack16<T>(v)); st16(C2 + o + 4, pack16<T>(v + 4)); }
              }
              if (epi.act == SPG_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
              } else if (epi.act == SPG_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
              }
              if (Hh) {
                float h[8];
                if constexpr (sizeof(T) == 2) unpack16<T>(ld16(Hh + o), h);
                else { unpack16<T>(ld16(Hh + o), h); unpack16<T>(ld16(Hh + o + 4), h + 4); }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(h[e]);
              }
              if (R) {
                float rr[8];
                if constexpr (sizeof(T) == 2) unpack16<T>(ld16(R + o), rr);
                else { unpack16<T>(ld16(R + o), rr); unpack16<T>(ld16(R + o + 4), rr + 4); }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rr[e];
              }
              if (DBG != 4 || M < 0) {  // DBG 4: whole epilogue except the output store
                if constexpr (sizeof(T) == 2) st16(C + o, pack16<T>(v));
                else { st16(C + o, pack16<T>(v)); st16(C + o + 4, pack16<T>(v + 4)); }
              } else {
                asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
              }
            } else {
              for (int e = 0; e < 8 && n + e < N; ++e) {
                float x = v[e];
                if (epi.bias) x += epi.bias[n + e];
                if (C2) ST<T>::st(C2 + o + e, x);
                if (epi.act == SPG_ACT_GELU) x = gelu_f(x);
                else if (epi.act == SPG_ACT_RELU) x = fmaxf(x, 0.f);
                if (Hh) x *= gelu_grad_f(ST<T>::ld(Hh + o + e));
                if (R) x += ST<T>::ld(R + o + e);
                ST<T>::st(C + o + e, x);
              }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      kt = 0; ++j;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_pipe_kernel: the default bf16 NT kernel.  8 waves (4 m x 2 n, wave tile 32 m x 16*NB n), persistent over tiles,
// LDS-DMA fill 2 steps ahead into a 4-stage ring.  What it adds over the plain DMA kernel above:
//   * fragments are double-buffered in registers: step k+1's ds_reads are issued in the shadows of step k's MFMAs;
//   * the step body is ONE basic block whose order is pinned group by group (an MFMA, then in its shadow a fragment read and a slice
//     of the DMA issue): both waves of a SIMD are phase-locked by the per-step barrier, so whatever is not inside an MFMA shadow
//     leaves the matrix pipe idle (measured: 0.54 -> 0.45 us per step with the fill and epilogue switched off);
//   * the epilogue is branch-free: activation mode is a template parameter; bias, residual, gelu_h and both outputs go through
//     bounds-checked buffer descriptors (zero-sized when the operand is absent: loads return 0, stores are dropped), and its operand
//     loads are issued at the top of the tile's last step, ahead of that step's DMA pieces, so the in-order vmcnt never makes them
//     wait for a fresh fill.
// (A deferred variant that parked the finished accumulators and drained them inside the next tile's MFMA shadows was measured and
// dropped: its steps ran 1.8-2.4x a plain step -- VGPR spills at NB = 4 and operand waits queued behind the previous DMA group.)
// LDS: 4 x 32 KiB stages + 8 x 4 KiB swizzled slabs = 160 KiB (one workgroup per CU).
// vmcnt bookkeeping: VMEM ops retire in order, so the wait for stage k+1 may leave outstanding exactly the younger ops: the
// 4 pieces of group k+2 plus the epilogue stores of the last two steps (every store instruction executes: lanes are masked by
// out-of-range offsets, not by EXEC).
// ------------------------------------------------------------------------------------------------
constexpr int PIPE_STAGES = 4;
constexpr int PIPE_SLAB_BYTES = 16 * 64 * 4;
constexpr int PIPE_LDS_BYTES = PIPE_STAGES * DMA_STAGE_BYTES + 8 * PIPE_SLAB_BYTES;
enum { PIPE_ACT_NONE = 0, PIPE_ACT_GELU = 1, PIPE_ACT_HH = 2,   // none | C2 = preact, C = gelu(.) | C = (.) * gelu'(gelu_h)
       PIPE_ACT_GELU_D = 3, PIPE_ACT_MULH = 4 };                   // C2 = gelu'(preact), C = gelu(.) | C = (.) * gelu_h

__device__ __forceinline__ void wait_vm(int n) {  // s_waitcnt vmcnt(<= n), n wave-uniform, n >= 8
  if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void bstore16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const u32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(bufvec_t{v.x, v.y, v.z, v.w}, r, byte_off, 0, 0);
}
template <int AUX> __device__ __forceinline__ void bstore16_aux(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const u32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(bufvec_t{v.x, v.y, v.z, v.w}, r, byte_off, 0, AUX);
}
template <int AUX> __device__ __forceinline__ u32x4 bload16_aux(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  bufvec_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
  return u32x4{v[0], v[1], v[2], v[3]};
}

template <typename T, int NB> struct NtFrags {
  typename Mma<T>::Frag a[Mma<T>::SUB][NB], b[Mma<T>::SUB][2];
};
template <typename T, int NB>
__device__ __forceinline__ void load_frags(const char* __restrict__ st, int wn, int wm, int lane, NtFrags<T, NB>& f) {
  using M_ = Mma<T>;
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < M_::SUB; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rowb = wm * 32 + i * 16 + r;
      f.b[s][i] = M_::load(st + rowb * ROWB, rowb & 7, s, q);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int rowa = wn * (16 * NB) + i * 16 + r;
      f.a[s][i] = M_::load(st + BM * ROWB + rowa * ROWB, rowa & 7, s, q);
    }
  }
}

struct PipeEpi {   // byte sizes of the optional epilogue operands (0 = absent)
  unsigned c_bytes, c2_bytes, r_bytes, h_bytes, bias_bytes;
  unsigned pf_bytes = 0;        // warm-up hint (spg_prefetch_hint): bytes at pf that the NEXT launch will read cold -- see prefetch_lines()
  const void* pf = nullptr;
};

// The warm-up hint of the next launch's weights, consumed by the spg_gemm_nt call that follows spg_prefetch_hint on the same thread.
static thread_local const void* t_hint_ptr = nullptr;
static thread_local unsigned t_hint_bytes = 0;
static thread_local const void* t_cur_pf = nullptr;       // ... and held here for the launchers while that call runs
static thread_local unsigned t_cur_pf_bytes = 0;

// One 4-byte LDS-DMA load per 64 bytes of [pf, pf + bytes), 4 KiB per wave instruction, issued before the wave's first real piece
// and never waited for by itself: nothing lands in a VGPR, the data goes to `scratch` (256 bytes of LDS that this same wave overwrites with
// a later, in-order DMA piece or writes before reading) and is never read.  The point is the side effect: the lines are in the Infinity
// Cache (and one XCD's L2) when the next launch asks for them.  In the step every bf16 weight matrix is read once per pass from HBM (862 MB
// of copies against a 256 MB cache), at the head of a launch whose X operand is warm: measured on a chain of 36 stage-3 blocks with their
// own weights (tools/prefetch_probe.py), weights touched one kernel ahead take 1.4-2.3 us off each GEMM (17.5 -> 15.3 us proj / fc2,
// 16.3 -> 14.5 qkv, 25.7 -> 24.2 fc1 + GELU).  vmcnt: the hint's loads are the wave's OLDEST outstanding memory operations, so every
// counted wait `vmcnt(n)` that follows still means "all but the n youngest have landed".
__device__ __forceinline__ void prefetch_lines(const void* pf, unsigned bytes, char* scratch, unsigned gwave, unsigned nwaves, int lane) {
  if (bytes == 0u) return;
  const __amdgpu_buffer_rsrc_t pr = make_rsrc(pf, bytes);
#ifndef SPG_PF_STRIDE     // bytes between a wave's touches (tools/ A/B builds: 128 -> 21.56 ms per step, 64 -> 21.50, 32 -> 21.47, no hint 21.94 / 21.8)
#define SPG_PF_STRIDE 64
#endif
#pragma unroll 1
  for (unsigned base = gwave * (64u * SPG_PF_STRIDE); base < bytes; base += nwaves * (64u * SPG_PF_STRIDE))
    __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (__attribute__((address_space(3))) void*)scratch, 4, base + (unsigned)lane * SPG_PF_STRIDE, 0, 0, 0);
}

// One problem of the persistent pipelined kernel, and the cross-workgroup dependencies of a phase of nt_chain_kernel (below): the body is
// shared by the plain kernel (one problem per launch, CH = 0: nothing of the chain machinery is compiled in) and the chain kernel.
struct PipeProb {
  const void* X; const void* W; void* C;
  NtEpi epi; PipeEpi pe;
  int M, N, K, ldx, ldc;
  int tiles_n, ntiles;
  unsigned xbytes, wbytes;
};
struct ChainSync {
  const unsigned* wait;     // CH & 1: per 128-row block, arrivals of the producing phase (tiles / row items that wrote rows of this block)
  unsigned wait_target;     //         ... that make the block complete
  unsigned* sig;            // CH & 2: per 128-row block, this phase's arrival counter (one arrival per finished tile)
  unsigned* err;            // set to 1 when a bounded wait gave up (the launch then finishes with wrong results instead of hanging)
};
constexpr int CHAIN_SPIN_LIMIT = 1 << 17;    // polls of ~0.5-1 us each before a wait gives up
constexpr int CH_SC1 = 16;                   // cache-policy bit sc1 of the buffer instructions (agent scope: write-through stores, L1-bypassing loads)

// Waits until row block `tm` of the producing phase is complete (every lane polls the same word; relaxed agent-scope loads, s_sleep
// between polls: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The data itself is read by
// sc1 loads afterwards (X through LDS-DMA, residual / gelu_h to registers), the producers stored it sc1 and drained their stores before
// their arrival.
__device__ __forceinline__ void chain_wait_block(const ChainSync& cs, int tm) {
  if (cs.wait == nullptr) return;          // (a chain's first phase: its rows were written before the launch)
  int spins = 0;
  while (__hip_atomic_load(cs.wait + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cs.wait_target) {
    if (++spins >= CHAIN_SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(cs.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

template <typename T, bool CONV, int ACT, int NB, bool DEFER = false, int DBG = 0, int CH = 0>
__device__ __forceinline__ void nt_pipe_body(const PipeProb& pb, const ChainSync& cs, const ConvGeom& g, char* __restrict__ smem,
                                             const int first, const int G) {
  static_assert(sizeof(T) == 2, "bf16 only");
  static_assert(CH == 0 || (!CONV && !DEFER), "chain phases are dense problems on the plain epilogue");
  const T* __restrict__ X = (const T*)pb.X;
  const T* __restrict__ W = (const T*)pb.W;
  T* __restrict__ C = (T*)pb.C;
  const NtEpi epi = pb.epi;
  const PipeEpi pe = pb.pe;
  const int M = pb.M, N = pb.N, K = pb.K, ldx = pb.ldx, ldc = pb.ldc, tiles_n = pb.tiles_n, ntiles = pb.ntiles;
  const unsigned xbytes = pb.xbytes, wbytes = pb.wbytes;
  constexpr int X_AUX = (CH & 1) ? CH_SC1 : 0;      // operands another workgroup of this launch wrote: L1-bypassing loads
#ifndef SPG_PIPE_C_AUX    // (tools/ A/B builds: cache policy of the plain persistent kernel's output stores, see SPG_V3_C_AUX)
#define SPG_PIPE_C_AUX 0
#endif
  constexpr int C_AUX = (CH & 2) ? CH_SC1 : SPG_PIPE_C_AUX;      // results another workgroup of this launch reads: write-through stores
  using M_ = Mma<T>;
  constexpr int VEC = ST<T>::VEC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int NP = 2;                  // DMA pieces (1 KiB) per operand per wave per stage
  // DEFER parks finished tiles in LDS (8 KiB per wave), paid for with one fill stage: 3 x 32 + 8 x 8 = 160 KiB as well
  constexpr int STAGES = DEFER ? PIPE_STAGES - 1 : PIPE_STAGES;
  constexpr int SLAB_BYTES = DEFER ? 2 * PIPE_SLAB_BYTES : PIPE_SLAB_BYTES;
  constexpr int BN_ = 32 * NB;
  constexpr bool NO_DMA = DBG == 1 || DBG == 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int nkt = (K + BK - 1) / BK;
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + G - 1) / G;
  const int total = my_tiles * nkt;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(X, xbytes), wr = make_rsrc(W, wbytes);
  const __amdgpu_buffer_rsrc_t cr = make_rsrc(C, pe.c_bytes), c2r = make_rsrc(epi.C2, pe.c2_bytes);
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(epi.residual, pe.r_bytes), hr = make_rsrc(epi.gelu_h, pe.h_bytes);
  const __amdgpu_buffer_rsrc_t br = make_rsrc(epi.bias, pe.bias_bytes);
  if constexpr (CH == 0 && !DEFER) {   // the next launch's weights: into this wave's epilogue slab (written before it is read, much later)
    prefetch_lines(pe.pf, pe.pf_bytes, smem + STAGES * DMA_STAGE_BYTES + wave * SLAB_BYTES, (unsigned)(blockIdx.x * 8 + wave), (unsigned)(G * 8), lane);
  }

  // ---- DMA issue stream (2 steps ahead of the fragment reads, 3 ahead of the MFMAs)
  const int lrow = lane >> 3, lp = lane & 7;
  int is_kt = 0, is_tile = first, is_m0, is_n0, is_slot = 0;
  bool is_live = true;     // false once the stream has run past this workgroup's last tile: pieces become OOB no-ops
  int py[NP], px[NP];
  auto enter_tile = [&]() __attribute__((always_inline)) {
    const int tn = is_tile % tiles_n, tm = is_tile / tiles_n;
    is_m0 = tm * BM; is_n0 = tn * BN_;
    if constexpr ((CH & 1) != 0) chain_wait_block(cs, tm);     // the rows this tile reads (X; residual rows follow transitively)
    if constexpr (CONV) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int m = is_m0 + 64 * i + wave * 8 + lrow;
        const int hw = g.H * g.W;
        const int b = m / hw, rem = m - b * hw;
        py[i] = rem / g.W; px[i] = rem - py[i] * g.W;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) { py[i] = 0; px[i] = 0; }
    }
  };
  enter_tile();
  // branch-free: every step issues exactly 2*NP pieces (a fixed vmcnt cost the waits rely on), split in address / go halves so
  // the step body can drop each half into a different MFMA shadow
  unsigned dxo[NP], dwo[NP];
  auto dma_addr = [&](int i) __attribute__((always_inline)) {
    const int row = 64 * i + wave * 8 + lrow;
    const int k0 = is_kt * BK + ((lp ^ (row & 7)) * VEC);
    const bool kin = is_live && k0 < K;
    const unsigned xo = x_chunk_off<T, CONV>(is_m0 + row, k0, ldx, g, py[i], px[i]);
    const unsigned wo = (unsigned)(((long)(is_n0 + row) * K + k0) * (long)sizeof(T));
    dxo[i] = kin ? xo : OOB;
    dwo[i] = (kin && row < BN_) ? wo : OOB;
  };
  auto dma_go = [&](int i, int which) __attribute__((always_inline)) {
    char* st = smem + is_slot * DMA_STAGE_BYTES;
    const int prow = 64 * i + wave * 8;
    if constexpr (!NO_DMA) {
      if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + prow * ROWB), 16, dxo[i], 0, 0, X_AUX);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + BM * ROWB + prow * ROWB), 16, dwo[i], 0, 0, 0);
    } else {
      asm volatile("" :: "v"(dxo[i]), "v"(dwo[i]));
    }
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    is_slot = is_slot + 1 == STAGES ? 0 : is_slot + 1;
    if (++is_kt == nkt) {
      is_kt = 0; is_tile += G;
      if (is_tile < ntiles) enter_tile(); else is_live = false;
    }
  };

  f32x4 acc[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float* slab = reinterpret_cast<float*>(smem + STAGES * DMA_STAGE_BYTES + wave * SLAB_BYTES);
  const int r15 = lane & 15, q = lane >> 4;
  const int erow = lane >> 3, ech = lane & 7;    // epilogue role: row within an 8-row half, 8-column group
  const bool ch_in = ech * 8 < 16 * NB;

  // ---- epilogue of the tile that ends with the current step: branch-free (absent operands have zero-sized descriptors, lanes are
  // masked by out-of-range offsets).  Its operand loads are issued at the top of the tile's last step, ahead of that step's DMA
  // pieces, so waiting for them never waits for a fresh fill.
  int tile = first, pm0 = 0, pn0 = 0;
  unsigned eo[2][2];              // [m block][row half]: byte offset of this lane's 8 outputs (OOB = masked)
  u32x4 er[2][2], eh[2][2];       // residual / gelu_h operands
  f32x4 eb0, eb1;                 // bias
  auto epi_request = [&]() __attribute__((always_inline)) {
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    pm0 = tm * BM; pn0 = tn * BN_;
    const int n = pn0 + wn * (16 * NB) + ech * 8;
    const bool nin = ch_in && n < N;           // N % 8 == 0 here: n < N <=> all 8 columns inside
    const unsigned nb_ = ch_in ? (unsigned)(n * 4) : OOB;
    const bufvec_t b0 = __builtin_amdgcn_raw_buffer_load_b128(br, nb_, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(br, nb_ + 16, 0, 0);
    eb0 = f32x4{__uint_as_float(b0[0]), __uint_as_float(b0[1]), __uint_as_float(b0[2]), __uint_as_float(b0[3])};
    eb1 = f32x4{__uint_as_float(b1[0]), __uint_as_float(b1[1]), __uint_as_float(b1[2]), __uint_as_float(b1[3])};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int m = pm0 + wm * 32 + qt * 16 + erow + 8 * jj;
        eo[qt][jj] = (nin && m < M) ? (unsigned)(((long)m * ldc + n) * 2) : OOB;
        er[qt][jj] = bload16_aux<X_AUX>(rr, eo[qt][jj]);
        if constexpr (ACT == PIPE_ACT_HH || ACT == PIPE_ACT_MULH) eh[qt][jj] = bload16(hr, eo[qt][jj]);
      }
  };
  auto epi_run = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      // transpose 16 rows through the wave's slab (LDS executes one wave's operations in order: no wait between write and read)
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) {
        *reinterpret_cast<f32x4*>(slab + r15 * 64 + (((ni * 4 + q) ^ r15) << 2)) = acc[ni][qt];
        acc[ni][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      f32x4 ea[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int row = erow + 8 * jj;
        ea[jj][0] = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * ech) ^ row) << 2));
        ea[jj][1] = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * ech + 1) ^ row) << 2));
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float ev[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ev[e] = ea[jj][0][e] + eb0[e]; ev[4 + e] = ea[jj][1][e] + eb1[e]; }
        if constexpr (ACT == PIPE_ACT_GELU) {
          if constexpr (DBG != 4) bstore16_aux<C_AUX>(c2r, eo[qt][jj], pack16<T>(ev));
#pragma unroll
          for (int e = 0; e < 8; ++e) ev[e] = gelu_f(ev[e]);
        }
        if constexpr (ACT == PIPE_ACT_GELU_D) {
          float dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_both_f(ev[e], ev[e], dv[e]);
          bstore16_aux<C_AUX>(c2r, eo[qt][jj], pack16<T>(dv));
        }
        if constexpr (ACT == PIPE_ACT_HH) {
          float h[8];
          unpack16<T>(eh[qt][jj], h);
#pragma unroll
          for (int e = 0; e < 8; ++e) ev[e] *= gelu_grad_f(h[e]);
        }
        if constexpr (ACT == PIPE_ACT_MULH) {
          float h[8];
          unpack16<T>(eh[qt][jj], h);
#pragma unroll
          for (int e = 0; e < 8; ++e) ev[e] *= h[e];
        }
        float rres[8];
        unpack16<T>(er[qt][jj], rres);
#pragma unroll
        for (int e = 0; e < 8; ++e) ev[e] += rres[e];
        if constexpr (DBG != 4) bstore16_aux<C_AUX>(cr, eo[qt][jj], pack16<T>(ev));
        else asm volatile("" :: "v"(ev[0]), "v"(ev[1]), "v"(ev[2]), "v"(ev[3]), "v"(ev[4]), "v"(ev[5]), "v"(ev[6]), "v"(ev[7]));
      }
    }
  };
  // DEFER: the same epilogue cut into pieces that ride in the MFMA shadows of the NEXT tile's first two steps (16 rows each).  The
  // pieces contain no VMEM loads (the operands were requested a tile ago), so nothing in them waits on the fill pipeline.
  f32x4 ca[2];
  float cv[8];
  auto chunk_piece = [&](auto QT_, int pc) __attribute__((always_inline)) {
    constexpr int qt = decltype(QT_)::value;
    if (pc == 2 || pc == 6) {          // read back one row half of the parked tile
      const int row = erow + 8 * (pc == 6);
      ca[0] = *reinterpret_cast<const f32x4*>(slab + (qt * 16 + row) * 64 + (((2 * ech) ^ row) << 2));
      ca[1] = *reinterpret_cast<const f32x4*>(slab + (qt * 16 + row) * 64 + (((2 * ech + 1) ^ row) << 2));
    } else if (pc == 3 || pc == 7) {
      const int jj = pc == 7;
#pragma unroll
      for (int e = 0; e < 4; ++e) { cv[e] = ca[0][e] + eb0[e]; cv[4 + e] = ca[1][e] + eb1[e]; }
      if constexpr (ACT == PIPE_ACT_GELU) {
        bstore16_aux<C_AUX>(c2r, eo[qt][jj], pack16<T>(cv));
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[e] = gelu_f(cv[e]);
      }
      if constexpr (ACT == PIPE_ACT_HH) {
        float h[8];
        unpack16<T>(eh[qt][jj], h);
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[e] *= gelu_grad_f(h[e]);
      }
    } else if (pc == 4 || pc == 8) {
      const int jj = pc == 8;
      if constexpr (ACT == PIPE_ACT_GELU) {
#pragma unroll
        for (int e = 4; e < 8; ++e) cv[e] = gelu_f(cv[e]);
      }
      if constexpr (ACT == PIPE_ACT_HH) {
        float h[8];
        unpack16<T>(eh[qt][jj], h);
#pragma unroll
        for (int e = 4; e < 8; ++e) cv[e] *= gelu_grad_f(h[e]);
      }
    } else if (pc == 5 || pc == 9) {
      const int jj = pc == 9;
      float rres[8];
      unpack16<T>(er[qt][jj], rres);
#pragma unroll
      for (int e = 0; e < 8; ++e) cv[e] += rres[e];
      bstore16_aux<C_AUX>(cr, eo[qt][jj], pack16<T>(cv));
    }
  };
  constexpr int stores_per_chunk = ((ACT == PIPE_ACT_GELU || ACT == PIPE_ACT_GELU_D) ? 4 : 2);
  constexpr int stores_per_tile = ((ACT == PIPE_ACT_GELU || ACT == PIPE_ACT_GELU_D) ? 8 : 4);

  // ---- prologue: 4 groups in flight, stage 0 landed, its fragments requested.  Because a stage's fragments sit in registers one
  // step before they are multiplied, its LDS slot is free again at the top of that step: group k+4 goes into stage k's slot, i.e.
  // three groups (96 KiB per CU) stay in flight behind the one being read.
  for (int i = 0; i < STAGES; ++i) {
#pragma unroll
    for (int j = 0; j < NP; ++j) { dma_addr(j); dma_go(j, 0); dma_go(j, 1); }
    issue_advance();
  }
  if constexpr (STAGES == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  NtFrags<T, NB> fa, fb;
  load_frags<T, NB>(smem, wn, wm, lane, fa);

  int kt = 0, rd_slot = 1;   // rd_slot: the stage whose fragments the current step reads ahead
  int pend = 0;              // DEFER: chunks of the parked tile still to drain (2 -> rows 0..15 next, 1 -> rows 16..31)
  int st1 = 0, st2 = 0;      // epilogue stores issued during the previous step and the one before
  // CH & 2: a finished tile's arrival.  Its epilogue stores (issued in the tile's last step s) are older than the eight youngest
  // operations the wait at the top of step s + 3 leaves outstanding (st1 = st2 = 0 there: tiles have >= 4 steps), so after that step's
  // barrier every wave's stores of the tile have completed (sc1: written through) and one lane adds 1 to the row block's counter.
  // (The waits themselves only get stricter by the extra operations -- polls, arrivals -- a chain phase puts into the queue.)
  int sig_cd = 0, sig_tm = 0;
  auto chain_signal = [&]() __attribute__((always_inline)) {
    if (tid == 0 && cs.sig != nullptr) __hip_atomic_fetch_add(cs.sig + sig_tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the pinned block: MFMA i, then in its shadow fragment read i of the next step and a slice of the DMA issue (addresses after
  // MFMAs 0 and 3, the four pieces after MFMAs 1, 2, 4, 5); on a tile's last step the epilogue operand requests ride in shadow 0
  auto mma_block = [&](NtFrags<T, NB>& cur, NtFrags<T, NB>& nxt, auto MODE_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(MODE_)::value;   // 0 plain step, 1 a tile's last step, 2 / 3 first / second step after a parked tile
    constexpr bool LAST = MODE == 1;
    constexpr int dma_mode = DBG == 8 ? 1 : (DBG == 9 ? 2 : 0);   // experiment: 1 = burst before the block, 2 = late in the block
    if constexpr (dma_mode == 1) {
      if constexpr (LAST) epi_request();
#pragma unroll
      for (int j = 0; j < NP; ++j) { dma_addr(j); dma_go(j, 0); dma_go(j, 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int NM = 2 * NB * 2, NR = 2 * (NB + 2);
    const char* rst = smem + rd_slot * DMA_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if constexpr (DBG != 2) {
        const int ms = i / (NB * 2), mr = i % (NB * 2), ni = mr >> 1, mi = mr & 1;
        acc[ni][mi] = M_::mma(cur.a[ms][ni], cur.b[ms][mi], acc[ni][mi]);
      }
      if constexpr (LAST && DBG != 3 && dma_mode != 1) {
        if (i == 0) epi_request();
      }
      if (i < NR) {
        const int rs = i / (NB + 2), rr_ = i % (NB + 2);
        if (rr_ < 2) {
          const int rowb = wm * 32 + rr_ * 16 + r15;
          nxt.b[rs][rr_] = M_::load(rst + rowb * ROWB, rowb & 7, rs, q);
        } else {
          const int rowa = wn * (16 * NB) + (rr_ - 2) * 16 + r15;
          nxt.a[rs][rr_ - 2] = M_::load(rst + BM * ROWB + rowa * ROWB, rowa & 7, rs, q);
        }
      }
      if (dma_mode == 0) {
        if (i == 0) dma_addr(0);
        if (i == 1) dma_go(0, 0);
        if (i == 2) dma_go(0, 1);
        if (i == 3) dma_addr(1);
        if (i == 4) dma_go(1, 0);
        if (i == 5) dma_go(1, 1);
        if (i == 6) issue_advance();   // (inside the block: scalar work in the issue slots the MFMAs leave free)
      } else if (dma_mode == 2) {   // late: behind the fragment reads
        if (i == NM - 6) dma_addr(0);
        if (i == NM - 5) dma_go(0, 0);
        if (i == NM - 4) dma_go(0, 1);
        if (i == NM - 3) dma_addr(1);
        if (i == NM - 2) dma_go(1, 0);
        if (i == NM - 1) dma_go(1, 1);
      }
      if constexpr (DEFER && MODE >= 2) {   // 10 chunk pieces spread over the block
#pragma unroll
        for (int pc = 0; pc < 10; ++pc)
          if (pc * NM / 10 == i) chunk_piece(std::integral_constant<int, MODE - 2>{}, pc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto step = [&](NtFrags<T, NB>& cur, NtFrags<T, NB>& nxt) __attribute__((always_inline)) {
    // stage gc+1 has landed (this wave's pieces): younger than it are groups gc+2 and gc+3 (8 ops) and the last two steps' epilogue
    // stores (STAGES - 2 groups in general).  lgkmcnt(0): this wave's reads of stage gc (issued during the previous step) have returned, so after the barrier
    // nobody still reads the slot that group gc+4 is about to overwrite.
    if constexpr (STAGES == 4) {
      if (st1 + st2 == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else { wait_vm(8 + st1 + st2); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    } else {   // 3 stages: one group (gc+2) younger than the awaited one
      if (st1 + st2 == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (st1 + st2 >= 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                       // ... and everyone's
    __builtin_amdgcn_sched_barrier(0);
    st2 = st1; st1 = 0;
    if constexpr ((CH & 2) != 0) {
      if (sig_cd > 0 && --sig_cd == 0) chain_signal();
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the counters are updated by value selects after the branch: symmetric "+=" in both arms gets merged into one store through a
    // selected pointer, which pins both variables in scratch memory and puts a vmcnt(0) reload at the top of every step)
    const bool last = __builtin_amdgcn_readfirstlane(kt + 1) == nkt;
    int ran_chunk = 0;
    if (DEFER && pend == 2) { mma_block(cur, nxt, std::integral_constant<int, 2>{}); ran_chunk = 1; }
    else if (DEFER && pend == 1) { mma_block(cur, nxt, std::integral_constant<int, 3>{}); ran_chunk = 1; }
    else if (last) {
      mma_block(cur, nxt, std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG == 3) {
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            if (M < 0) *reinterpret_cast<f32x4*>(slab + r15 * 64 + ni * 4) = acc[ni][mi];
            acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      } else if constexpr (DEFER) {   // park in the wave's slab, transposed on the way (tiles have >= 3 steps here: the two chunk
                                      // steps never coincide with a last step, and the slab is long drained before the next park)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            *reinterpret_cast<f32x4*>(slab + (mi * 16 + r15) * 64 + (((ni * 4 + q) ^ r15) << 2)) = acc[ni][mi];
            acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      } else {
        epi_run();
        st1 = stores_per_tile;
        if constexpr ((CH & 2) != 0) { sig_cd = 3; sig_tm = pm0 / BM; }
      }
    } else {
      mma_block(cur, nxt, std::integral_constant<int, 0>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DEFER) {
      st1 = ran_chunk ? stores_per_chunk : st1;
      pend = ran_chunk ? pend - 1 : (last && DBG != 3 ? 2 : pend);
    }
    kt = last ? 0 : kt + 1;
    tile = last ? tile + G : tile;
    if constexpr (DBG == 8 || DBG == 9) issue_advance();   // (the ablation modes issue their DMA elsewhere in the block)
    rd_slot = rd_slot + 1 == STAGES ? 0 : rd_slot + 1;
  };
  for (int gc = 0; gc < total; gc += 2) {
    step(fa, fb);
    if (gc + 1 < total) step(fb, fa);
  }
  if constexpr (DEFER) {   // the last tile's parked accumulators
    if (pend == 2) {
#pragma unroll
      for (int pc = 0; pc < 10; ++pc) chunk_piece(std::integral_constant<int, 0>{}, pc);
      pend = 1;
    }
    if (pend == 1) {
#pragma unroll
      for (int pc = 0; pc < 10; ++pc) chunk_piece(std::integral_constant<int, 1>{}, pc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing no-op pieces
  if constexpr ((CH & 2) != 0) {                      // the last tile's arrival (and, at the phase seam, every store of this workgroup is out)
    __builtin_amdgcn_s_barrier();
    if (sig_cd > 0) chain_signal();
  }
}

template <typename T, bool CONV, int ACT, int NB, bool DEFER = false, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_nt_pipe_kernel(const T* __restrict__ X, const T* __restrict__ W, T* __restrict__ C,
                                                            NtEpi epi, PipeEpi pe, int M, int N, int K, int ldx, int ldc,
                                                            ConvGeom g, int tiles_n, int ntiles, unsigned xbytes, unsigned wbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PipeProb pb{X, W, C, epi, pe, M, N, K, ldx, ldc, tiles_n, ntiles, xbytes, wbytes};
  const ChainSync cs{nullptr, 0u, nullptr, nullptr};
  const int G = (int)gridDim.x;
  nt_pipe_body<T, CONV, ACT, NB, DEFER, DBG, 0>(pb, cs, g, smem, xcd_remap(blockIdx.x, G), G);
}

#ifdef SPG_DEV_KERNELS
#include "dev/nt_chain_kernel.inc"
#endif

// LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS).  A template (on the pointer type) so that the device-only 16-byte form of the builtin is
// checked only at instantiation: in a non-dependent statement hipcc's host pass fails the check silently and drops the kernel's stub.
template <typename T>
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, T* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_v3: 4 waves (2 x 2), ONE 128 x (32 NB) tile per workgroup, TWO workgroups per CU.
//
// Why (measured on the persistent 8-wave kernel above, profiles/round2): its 32 x 16NB wave tile needs 0.75 ds_read_b128 per MFMA and
// the LDS-DMA fill another 0.25, i.e. the LDS array (256 B/clk/CU, one b128 read per 16-cycle MFMA slot and SIMD) is saturated at
// ~53 % MFMA occupancy, and the epilogue (~2.3 us per tile, 40 % of a K = 576 tile) runs with every wave of the CU out of the MFMA
// pipe.  Here
//   * a wave owns 64 (m) x 16 NB (n): (4 + NB) / (4 NB) = 0.5 reads per MFMA at NB = 4 (+ 0.25 fill);
//   * a workgroup is 4 waves and NS x 16 KiB of LDS (NS = 4: 64 KiB), so two are resident per CU (2 waves per SIMD from DIFFERENT
//     tiles): one tile's prologue / epilogue / barrier waits sit under the other tile's MFMAs, and the hardware dispatcher -- not a
//     static persistent schedule -- balances tiles over CUs (grids that coexist with an RCCL kernel need no CU budget either);
//   * K advances 32 per step (64-byte LDS rows, one MFMA k-block per stage) so that NS - 1 = 3 DMA groups (48 KiB per workgroup,
//     96 KiB per CU) stay in flight inside the 80 KiB a workgroup may hold;
//   * the epilogue needs no LDS transpose: W rows are PERMUTED on their way into LDS (row 16 ni + 4 q + e of a wave's W slice holds
//     W[n0 + 32 (ni >> 1) + 8 q + 4 (ni & 1) + e]), so the accumulators (ni, ni + 1) of one m block hold 8 CONSECUTIVE output columns
//     of one row: bias / residual / gelu_h are read and C / C2 written as 16-byte vectors straight from the accumulators, four
//     lanes covering 64 contiguous bytes of a row.
// LDS image: 64-byte rows; 16-byte chunk c of row r holds logical chunk c ^ (2 * ((r >> 3) & 1)) -- with the fixed lane groups of
// ds_read_b128 ({0-3, 12-15, 20-27}, ...) the 16 rows x 4 chunks of a fragment read then fall on 16 distinct 4-bank groups.
// Pipeline (step t = 32 of K): frags(t) are in registers; barrier; MFMAs(t) | ds_read frags(t+1) from slot (t+1) % NS | DMA group
// t + NS into slot t % NS (whose fragments everybody has read: lgkmcnt(0) before the barrier).  Group t+1 must have landed at the
// barrier: younger than it are NS - 2 groups, so the wait is vmcnt((NS - 2) * pieces); once the groups run out, vmcnt(0).
// ------------------------------------------------------------------------------------------------
template <int N_> __device__ __forceinline__ void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory"); }
template <int N_> __device__ __forceinline__ void wait_vm_only() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// KS = MFMA k-blocks (32 of K) per LDS stage: 2 -> 128-byte rows, chunk c of row r holds logical chunk c ^ (r & 7) (the image of the
// kernels above); 1 -> 64-byte rows, chunk c of row r holds c ^ (2 ((r >> 3) & 1)).  NS = stages.  (KS, NS) = (2, 2): 64 KiB, one DMA
// group (32 KiB) in flight behind the stage being read; (1, 4): 64 KiB, three 16 KiB groups in flight, a barrier every 32 of K.
// Cache policy of the output stores (tools/ A/B builds: 16 = sc1, write-through, the line does not stay in the XCD's L2; 2 = nt).  Measured
// (DESIGN.md 3.1 item 49): sc1 takes 10-15 % off a 2304-wide launch REPLAYED back to back (fc1 23.2 -> 20.7 us, qkv 14.9 -> 12.9), changes
// nothing in the batch-8 train step and costs the batch-64 inference forward 4 % (1575 -> 1513 img/s): plain stores stay.
#ifndef SPG_V3_C_AUX
#define SPG_V3_C_AUX 0
#endif
#ifndef SPG_V3_C2_AUX
#define SPG_V3_C2_AUX 0
#endif
#ifndef SPG_V3_H_AUX
#define SPG_V3_H_AUX 0
#endif
template <bool CONV, int ACT, int NB, int KS, int NS>
__global__ __launch_bounds__(256, 2) void gemm_nt_v3_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                            NtEpi epi, PipeEpi pe, int M, int N, int K, int ldx, int ldc, ConvGeom g,
                                                            int tiles_n, unsigned xbytes, unsigned wbytes) {
  using T = bf16_t;
  static_assert(NB == 4 || NB == 2, "accumulator pairs hold 8 consecutive columns");
  static_assert(KS == 1 || KS == 2, "");
  constexpr int RB = 64 * KS;               // LDS row bytes
  constexpr int PR = 1024 / RB;             // rows per 1 KiB DMA piece
  constexpr int BN_ = 32 * NB;
  constexpr int STAGE = (BM + BN_) * RB;
  constexpr int WROWS = BN_ / 4;            // W rows each wave fills per stage
  constexpr int XP = 32 / PR, WP = WROWS / PR;
  constexpr int NP = XP + WP;               // DMA pieces per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int nh = (K + 31) >> 5;             // 32-deep steps
  const int nst = (nh + KS - 1) / KS;       // stages (DMA groups)
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN_;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(X, xbytes), wr = make_rsrc(W, wbytes);
  // the next launch's weights (prefetch_lines): into the first 256 bytes of the 1 KiB this wave's own first X piece of stage 0 fills next
  prefetch_lines(pe.pf, pe.pf_bytes, smem + (wave * 32) * RB, (unsigned)(blockIdx.x * 4 + wave), (unsigned)(gridDim.x * 4), lane);

  // ---- fill roles: wave w moves X rows [32 w, 32 w + 32) and W rows [WROWS w, WROWS (w + 1)) of the stage image
  const int lrow = KS == 2 ? lane >> 3 : lane >> 2;                                       // row within a piece
  const int kch = KS == 2 ? ((lane & 7) ^ lrow) : ((lane & 3) ^ ((lane >> 5) << 1));     // logical 16-byte chunk this lane fetches
  unsigned xoff[XP], woff[WP];
  bool xin[XP], win[WP];
  int py[XP], px[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    const int m = m0 + wave * 32 + p * PR + lrow;
    xin[p] = m < M;
    if constexpr (CONV) {
      const int hw = g.H * g.W;
      const int b = m / hw, rem = m - b * hw;
      py[p] = rem / g.W; px[p] = rem - py[p] * g.W;
      xoff[p] = (unsigned)m;
    } else {
      py[p] = 0; px[p] = 0;
      xoff[p] = (unsigned)(((long)m * ldx + kch * 8) * 2);
    }
  }
#pragma unroll
  for (int p = 0; p < WP; ++p) {
    const int i = wave * WROWS + p * PR + lrow;             // row of the W image
    const int half = i / (16 * NB), l = i - half * (16 * NB), ni = l >> 4, j = l & 15;
    const int n = n0 + half * (16 * NB) + (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);
    win[p] = n < N;
    woff[p] = (unsigned)(((long)n * K + kch * 8) * 2);
  }
  auto dma_piece = [&](int p, int gi, char* st) __attribute__((always_inline)) {   // piece p of DMA group (stage) gi
    const int k0 = gi * (32 * KS) + kch * 8;
    const bool kin = k0 < K;
    if (p < XP) {
      unsigned off;
      if constexpr (CONV) off = x_chunk_off<T, true>((int)xoff[p], k0, ldx, g, py[p], px[p]);
      else off = xoff[p] + (unsigned)gi * (unsigned)RB;
      lds_dma16(xr, st + (wave * 32 + p * PR) * RB, (kin && xin[p]) ? off : OOB);
    } else {
      const int q_ = p - XP;
      lds_dma16(wr, st + (BM + wave * WROWS + q_ * PR) * RB, (kin && win[q_]) ? woff[q_] + (unsigned)gi * (unsigned)RB : OOB);
    }
  };

  f32x4 acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r15 = lane & 15, q = lane >> 4;
  // this lane's fragment chunk inside a 16-row block, for k-block 0 (k-block 1 of a 128-byte row: the same ^ 64)
  const int fro = KS == 2 ? r15 * RB + ((q ^ (r15 & 7)) << 4) : r15 * RB + ((q ^ ((r15 >> 3) << 1)) << 4);
  const int boff = (wm * 64) * RB + fro, aoff = (BM + wn * (16 * NB)) * RB + fro;
  bf16x8_t fa0[NB], fb0[4], fa1[NB], fb1[4];
  auto read_frag = [&](bf16x8_t* fa, bf16x8_t* fb, const char* st, int sub, int r) __attribute__((always_inline)) {
    const int x = sub * 64;      // (only KS == 2 has sub == 1)
    if (r < 4) fb[r] = *reinterpret_cast<const bf16x8_t*>(st + ((boff + r * 16 * RB) ^ x));   // order of first use: m blocks, then n blocks
    else fa[r - 4] = *reinterpret_cast<const bf16x8_t*>(st + ((aoff + (r - 4) * 16 * RB) ^ x));
  };
  constexpr int NH = NB * 4, NR = NB + 4;   // MFMAs / fragment reads per step

  // ---- prologue: groups 0 .. NS-1 in flight, group 0 landed, its k-block 0 fragments requested
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    if (i < nst) {
#pragma unroll
      for (int p = 0; p < NP; ++p) dma_piece(p, i, smem + i * STAGE);
    }
  }
  if (nst >= NS) wait_vm_only<(NS - 1) * NP>();
  else wait_vm_only<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < NR; ++r) read_frag(fa0, fb0, smem, 0, r);

  // ---- epilogue operands (requested in the last step, after the last DMA group has landed)
  const int ncol = n0 + wn * (16 * NB) + q * 8;              // first of this lane's 8 consecutive columns of vector 0 (vector v: + 32 v)
  const __amdgpu_buffer_rsrc_t cr = make_rsrc(C, pe.c_bytes), c2r = make_rsrc(epi.C2, pe.c2_bytes);
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(epi.residual, pe.r_bytes), hr = make_rsrc(epi.gelu_h, pe.h_bytes);
  const __amdgpu_buffer_rsrc_t br = make_rsrc(epi.bias, pe.bias_bytes);
  constexpr int NV = NB / 2;                                 // 16-byte vectors per lane and m block
  unsigned eo[4];
  u32x4 er[4][NV], eh[4][NV];
  f32x4 eb[NB];
  auto epi_request = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bufvec_t b = __builtin_amdgcn_raw_buffer_load_b128(br, (unsigned)((ncol + 32 * (i >> 1) + 4 * (i & 1)) * 4), 0, 0);
      eb[i] = f32x4{__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + wm * 64 + mi * 16 + r15;
      eo[mi] = (m < M) ? (unsigned)(((long)m * ldc + ncol) * 2) : OOB;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const unsigned o = (eo[mi] != OOB && ncol + 32 * v < N) ? eo[mi] + 64u * v : OOB;
        er[mi][v] = bload16(rr, o);
        if constexpr (ACT == PIPE_ACT_HH || ACT == PIPE_ACT_MULH) eh[mi][v] = bload16_aux<SPG_V3_H_AUX>(hr, o);
      }
    }
  };

  // step t (32 of K): MFMAs from (ca, cb) | fragment reads of step t+1 into (na, nb).  BAR: step t+1 starts a new stage -- stage
  // gdone = (t + 1) / KS - 1 has been read completely, so after the barrier its slot takes DMA group gdone + NS, and group gdone + 1
  // must have landed: NS - 2 groups are younger than it (none once the groups have run out: vmcnt(0)).
  auto step = [&](bf16x8_t* ca, bf16x8_t* cb, bf16x8_t* na, bf16x8_t* nb, int t, auto BAR_) __attribute__((always_inline)) {
    constexpr bool BAR = decltype(BAR_)::value;
    const int gdone = (t + 1) / KS - 1;
    if constexpr (BAR) {
      if (gdone + NS <= nst) wait_vm_lgkm0<(NS - 2) * NP>();
      else wait_vm_lgkm0<0>();
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    const int rs = ((t + 1) / KS) % NS;
    const char* rst = smem + rs * STAGE;
    char* wst = smem + (BAR ? (gdone % NS) : 0) * STAGE;
    const bool fill = BAR && gdone + NS < nst;
    const int sub = KS == 2 ? ((t + 1) & 1) : 0;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      acc[i >> 2][i & 3] = Mma<T>::mma(ca[i >> 2], cb[i & 3], acc[i >> 2][i & 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (i < NR) read_frag(na, nb, rst, sub, i);
      if constexpr (BAR) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          constexpr int first = NP * 2 <= NH ? NH - 2 * NP + 1 : (NH > NP ? NH - NP : 0);   // every other slot of the tail, or every slot
          constexpr int stride = NP * 2 <= NH ? 2 : 1;
          if (first + stride * p == i && fill) dma_piece(p, gdone + NS, wst);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto last_step = [&](bf16x8_t* ca, bf16x8_t* cb) __attribute__((always_inline)) {
    epi_request();
#pragma unroll
    for (int i = 0; i < NH; ++i) acc[i >> 2][i & 3] = Mma<T>::mma(ca[i >> 2], cb[i & 3], acc[i >> 2][i & 3]);
  };
  for (int t = 0;;) {   // (t is even at the top: with KS == 2 only the second step of a pair crosses into a new stage)
    if (t + 1 >= nh) { last_step(fa0, fb0); break; }
    step(fa0, fb0, fa1, fb1, t, std::integral_constant<bool, KS == 1>{}); ++t;
    if (t + 1 >= nh) { last_step(fa1, fb1); break; }
    step(fa1, fb1, fa0, fb0, t, std::true_type{}); ++t;
  }

  // ---- epilogue: straight from the accumulators
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float ev[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ev[e] = acc[2 * v][mi][e] + eb[2 * v][e];
        ev[4 + e] = acc[2 * v + 1][mi][e] + eb[2 * v + 1][e];
      }
      const unsigned o = (eo[mi] != OOB && ncol + 32 * v < N) ? eo[mi] + 64u * v : OOB;
      if constexpr (ACT == PIPE_ACT_GELU) {
        bstore16_aux<SPG_V3_C2_AUX>(c2r, o, pack16<T>(ev));
#pragma unroll
        for (int e = 0; e < 8; ++e) ev[e] = gelu_f(ev[e]);
      }
      if constexpr (ACT == PIPE_ACT_GELU_D) {
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_both_f(ev[e], ev[e], dv[e]);
        bstore16_aux<SPG_V3_C2_AUX>(c2r, o, pack16<T>(dv));
      }
      if constexpr (ACT == PIPE_ACT_HH) {
        float h[8];
        unpack16<T>(eh[mi][v], h);
#pragma unroll
        for (int e = 0; e < 8; ++e) ev[e] *= gelu_grad_f(h[e]);
      }
      if constexpr (ACT == PIPE_ACT_MULH) {
        float h[8];
        unpack16<T>(eh[mi][v], h);
#pragma unroll
        for (int e = 0; e < 8; ++e) ev[e] *= h[e];
      }
      float rres[8];
      unpack16<T>(er[mi][v], rres);
#pragma unroll
      for (int e = 0; e < 8; ++e) ev[e] += rres[e];
      bstore16_aux<SPG_V3_C_AUX>(cr, o, pack16<T>(ev));
    }
  }
}

#ifdef SPG_DEV_KERNELS
#include "dev/gemm_nt_v5.inc"
#endif  // SPG_DEV_KERNELS (gemm_nt_v5)

// TN: dW[n][k] += sum_m dY[m][n] * X[m][k].  LDS rows are output features (n for the dY operand, k for the
// X operand), 128 bytes of consecutive m per row; register transpose of 4(m) x 16-byte patches.
// swizzle for these images: sw(f) = (f ^ (f >> 4)) & 7 -> fragment reads conflict free, patch writes 2-way.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_sw(int f) { return (f ^ (f >> 4)) & 7; }

template <typename T, int KI = 4>
__device__ __forceinline__ void mma_tile_tn(const char* __restrict__ As, const char* __restrict__ Bs, int wn, int wk,
                                            int lane, f32x4 (&acc)[4][KI]) {
  using M_ = Mma<T>;
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < M_::SUB; ++s) {
    typename M_::Frag a[4], b[KI];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowa = wn * 64 + i * 16 + r;
      a[i] = M_::load(As + rowa * ROWB, tn_sw(rowa), s, q);
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int rowb = wk * (16 * KI) + i * 16 + r;
      b[i] = M_::load(Bs + rowb * ROWB, tn_sw(rowb), s, q);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ki = 0; ki < KI; ++ki) acc[ni][ki] = M_::mma(a[ni], b[ki], acc[ni][ki]);
  }
}

// write a 4(m) x VEC(features) patch transposed into the LDS image. rows[4] = the four 16-byte loads.
template <typename T>
__device__ __forceinline__ void store_patch(char* img, const u32x4 (&rows)[4], int f0, int mg) {
  if constexpr (sizeof(T) == 2) {
    // 8 features; feature j gets the 4 bf16 (m = 4mg..4mg+3) = 8 bytes at chunk mg>>1, half mg&1
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = f0 + j;
      unsigned e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned word = rows[i][j >> 1];
        e[i] = (j & 1) ? (word >> 16) : (word & 0xffffu);
      }
      u32x2 v = {e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
      *reinterpret_cast<u32x2*>(img + f * ROWB + ((((mg >> 1) ^ tn_sw(f))) << 4) + ((mg & 1) << 3)) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + j;
      u32x4 v = {rows[0][j], rows[1][j], rows[2][j], rows[3][j]};
      *reinterpret_cast<u32x4*>(img + f * ROWB + ((mg ^ tn_sw(f)) << 4)) = v;
    }
  }
}

// WK = waves along k (2: 4 waves, each thread stages a dY patch AND an X patch; 4: 8 waves, threads 0-255 stage dY,
// 256-511 stage X, wave tile 64 n x 32 k)
template <typename T, bool CONV, int WK = 2>
__global__ __launch_bounds__(128 * WK) void gemm_tn_kernel(const T* __restrict__ dY, const T* __restrict__ X,
                                                             float* __restrict__ dW, int M, int N, int K, int ldy,
                                                             int ldx, int ldw, ConvGeom g, int tiles_k, int m_per_split,
                                                             unsigned ybytes, unsigned xbytes, float* __restrict__ dbias,
                                                             float* __restrict__ slabs) {
  constexpr int VEC = ST<T>::VEC;
  constexpr int MSTEP = ROWB / (int)sizeof(T);  // m rows per LDS tile: 64 (bf16) / 32 (f32)
  constexpr int NCH = 128 / VEC;                // feature chunks per tile row: 16 / 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                   // [2][128 n][ROWB]   (dY^T)
  char* Bs = smem + 2 * 128 * ROWB;  // [2][128 k][ROWB]   (X^T)
  constexpr int KI = 8 / WK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int stid = tid & 255;                  // staging thread id within its operand team
  const bool team_y = WK == 2 || tid < 256;    // stages dY patches
  const bool team_x = WK == 2 || tid >= 256;   // stages X patches
  // XCD-aware order: workgroups that share blockIdx%8 (one XCD, one L2) take consecutive (split, tile) units, i.e. the k-tiles
  // (for a 3x3 conv: the 9 taps) that re-read the same dY / X rows -- otherwise every tap re-fetches X from HBM.
  const int ntile = (int)gridDim.x;
  const int unit = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), ntile * (int)gridDim.y);
  const int tile_id = unit % ntile, split_id = unit / ntile;
  const int tk = tile_id % tiles_k, tn = tile_id / tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int m_begin = split_id * m_per_split;
  const int m_end = min(M, m_begin + m_per_split);

  const int fc = stid % NCH, mg = stid / NCH;  // feature chunk, m-group (4 rows each)

  f32x4 acc[4][KI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < KI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  // fused bias gradient: the k-tile-0 blocks also accumulate column sums of their dY patches (db[n] = sum_m dY[m][n])
  const bool do_bias = (dbias != nullptr) && (tk == 0) && team_y;
  float bsum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bsum[e] = 0.f;
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(dY, ybytes), xr = make_rsrc(X, xbytes);
  int tap_dy = 0, tap_dx = 0, tap_ci = 0;
  {
    const int k = k0 + fc * VEC;
    if constexpr (CONV) {
      const int tap = k / g.Ci;
      tap_ci = k - tap * g.Ci; tap_dy = tap / 3 - 1; tap_dx = tap - (tap / 3) * 3 - 1;
    }
  }
  auto gload = [&](int mt) {
    const int mb = m_begin + mt * MSTEP + mg * 4;
    const int n = n0 + fc * VEC, k = k0 + fc * VEC;
    const bool nin = n < N, kin = k < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + i;
      const bool mv = m < m_end;
      const unsigned yo = (unsigned)(((long)m * ldy + n) * (long)sizeof(T));
      if (team_y) ra[i] = bload16(yr, (mv && nin) ? yo : OOB);
      if (!team_x) continue;
      if constexpr (!CONV) {
        const unsigned xo = (unsigned)(((long)m * ldx + k) * (long)sizeof(T));
        rb[i] = bload16(xr, (mv && kin) ? xo : OOB);
      } else {
        const int hw = g.H * g.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / g.W, x = rem - y * g.W;
        const bool in = mv && (unsigned)(y + tap_dy) < (unsigned)g.H && (unsigned)(x + tap_dx) < (unsigned)g.W;
        const unsigned xo = (unsigned)((((long)m + (long)tap_dy * g.W + tap_dx) * g.Ci + tap_ci) * (long)sizeof(T));
        rb[i] = bload16(xr, (in && kin) ? xo : OOB);
      }
    }
  };
  auto sstore = [&](int buf) {
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[VEC];
        unpack16<T>(ra[i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) bsum[e] += f[e];
      }
    }
    if (team_y) store_patch<T>(As + buf * 128 * ROWB, ra, fc * VEC, mg);
    if (team_x) store_patch<T>(Bs + buf * 128 * ROWB, rb, fc * VEC, mg);
  };

  const int nm = (m_end - m_begin + MSTEP - 1) / MSTEP;
  if (nm <= 0) return;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int mt = 0; mt < nm; ++mt) {
    const int cur = mt & 1;
    if (mt + 1 < nm) gload(mt + 1);
    mma_tile_tn<T, KI>(As + cur * 128 * ROWB, Bs + cur * 128 * ROWB, wn, wk, lane, acc);
    if (mt + 1 < nm) sstore(cur ^ 1);
    __syncthreads();
  }
  if ((dbias != nullptr) && (tk == 0)) {  // reduce the per-thread partial sums over the m-groups through LDS (MFMA loop done)
    float* red = reinterpret_cast<float*>(smem);
    constexpr int NMG = 256 / NCH;
    if (team_y) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[mg * 128 + fc * VEC + e] = bsum[e];
    }
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
      for (int r = 0; r < NMG; ++r) t += red[r * 128 + tid];
      if (n0 + tid < N) atomicAdd(dbias + n0 + tid, t);
    }
  }
  // lane holds n = nb + 4q + r, k = kb + (lane&15)
  const int r15 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      const int k = k0 + wk * (16 * KI) + ki * 16 + r15;
      if (k >= K) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 64 + ni * 16 + q * 4 + e;
        if (n < N) {
          // split-M partials: plain stores into this split's slab (full store bandwidth; a reduce kernel folds the slabs into
          // the gradient) instead of f32 atomics (chip-wide ~1.3 TB/s: MI355X_MICROARCH.md "Global float atomics")
          if (slabs) slabs[((long)split_id * N + n) * K + k] = acc[ni][ki][e];
          else atomicAdd(dW + (long)n * ldw + k, acc[ni][ki][e]);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Transpose-read fragments for the LDS-DMA weight-gradient kernels below (gemm_tn_pipe_kernel, gemm_tn_group_kernel, ...).
// Tiles are DMA'd ROW-MAJOR ([m][128 features], 64 m rows of 256 B for bf16 / 32 rows of 512 B for f32) and the MFMA
// fragments (which need 8 consecutive m for one feature) come from the hardware transpose read ds_read_b64_tr_b16
// (bf16) or plain 4-byte reads (f32).  32-byte segments of a row are XOR-swizzled by row (on the DMA source side) so the
// four rows a transpose read touches sit on different banks.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int tnd_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T> struct TnFrag;
template <> struct TnFrag<bf16_t> {
  static constexpr int SUB = 2;                   // 32-row MFMA k-steps per 64-row tile
  using Frag = bf16x8_t;
  // feature block col0 (multiple of 16) of rows [32*s, 32*s+32) of a [64][128] bf16 tile
  __device__ static __forceinline__ Frag load(const char* tile, int s, int col0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int row0 = 32 * s + 8 * g + q;
    const int c = (col0 >> 3) + (p >> 1);         // logical 16-byte chunk
    const int within = (p & 1) * 8;
    const int o0 = row0 * 256 + ((((c >> 1) ^ tnd_swz(row0)) << 1 | (c & 1)) << 4) + within;
    const int row1 = row0 + 4;
    const int o1 = row1 * 256 + ((((c >> 1) ^ tnd_swz(row1)) << 1 | (c & 1)) << 4) + within;
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + o0));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
  __device__ static __forceinline__ Frag ones() {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    return __builtin_bit_cast(bf16x8_t, v);
  }
  // loop-invariant per-lane byte offsets of the two transpose reads of a fragment (precomputed once per kernel)
  __device__ static __forceinline__ void offsets(int s, int col0, int lane, int& o0, int& o1) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int row0 = 32 * s + 8 * g + q, row1 = row0 + 4;
    const int c = (col0 >> 3) + (p >> 1);
    const int within = (p & 1) * 8;
    o0 = row0 * 256 + ((((c >> 1) ^ tnd_swz(row0)) << 1 | (c & 1)) << 4) + within;
    o1 = row1 * 256 + ((((c >> 1) ^ tnd_swz(row1)) << 1 | (c & 1)) << 4) + within;
  }
  // one transpose read (half a fragment: 4 of its 8 rows) and the join of two halves
  __device__ static __forceinline__ s16x4_t load_half(const char* tile, int o) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + o));
  }
  __device__ static __forceinline__ Frag join(s16x4_t v0, s16x4_t v1) {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
  __device__ static __forceinline__ Frag load_at(const char* tile, int o0, int o1) {
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + o0));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
};
template <> struct TnFrag<float> {
  static constexpr int SUB = 8;                   // 4-row k-steps per 32-row tile
  using Frag = float;
  __device__ static __forceinline__ Frag load(const char* tile, int s, int col0, int lane) {
    return *reinterpret_cast<const float*>(tile + (4 * s + (lane >> 4)) * 512 + (col0 + (lane & 15)) * 4);
  }
  __device__ static __forceinline__ Frag ones() { return 1.f; }
  __device__ static __forceinline__ void offsets(int s, int col0, int lane, int& o0, int& o1) {
    o0 = (4 * s + (lane >> 4)) * 512 + (col0 + (lane & 15)) * 4; o1 = 0;
  }
  __device__ static __forceinline__ Frag load_at(const char* tile, int o0, int) { return *reinterpret_cast<const float*>(tile + o0); }
};

// ------------------------------------------------------------------------------------------------
// gemm_tn_pipe_kernel (bf16, dense X): the weight-gradient GEMM with the recipe of gemm_nt_pipe_kernel.
//   slab[split][n][k] = sum_{m in split} dY[m][n] * X[m][k]         (a reduce kernel folds the slabs into dW)
// Persistent over (128 x 128 tile, M split) units, 8 waves (2 n x 4 k, wave tile 64 n x 32 k).  Row-major [64 m][128] tiles of dY and
// X stream in by LDS-DMA (4-stage ring, three groups in flight, offsets advanced incrementally: no divisions in the loop); MFMA
// fragments come from ds_read_b64_tr_b16 transpose reads (no register transposes), double-buffered in registers one step ahead; the
// step body is one basic block with the order pinned per group (MFMA, then in its shadow two transpose reads and a slice of the DMA
// issue).  MFMA operand roles are (X fragment, dY fragment), so a lane owns 4 consecutive k of one n: the unit's result leaves as
// 16-byte stores.  The bias gradient (column sums of dY) is taken from the dY fragments already in registers with v_dot2_f32_bf16
// against ones, one n block per k-wave so the extra VALU work is spread evenly and nobody issues extra MFMAs.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// floor(m / d) for 0 <= m < 2^24 from a float reciprocal, corrected to exact (the conv gather's per-step pixel coordinates)
__device__ __forceinline__ int div_small(int m, int d, float rd) {
  int qq = (int)((float)m * rd);
  const int r = m - qq * d;
  qq += (r >= d) ? 1 : 0;
  qq -= (r < 0) ? 1 : 0;
  return qq;
}

#ifdef SPG_DEV_KERNELS
#include "dev/gemm_tn_pipe8.inc"
#endif  // SPG_DEV_KERNELS (8-wave single-problem wgrad kernel)

// ------------------------------------------------------------------------------------------------
// gemm_tn_pipe4_kernel: gemm_tn_pipe_kernel's schedule (units = tile x M split, partial sums to the split's slab) with SPECIALISED
// waves -- waves 0-3 multiply a 64 x 64 quarter of the tile each, waves 4-7 only issue the LDS-DMA fill (and, for the 3x3
// convolution, the gather arithmetic); see gemm_tn_group4_kernel for the measurements behind the split and for the hand-off.
// ------------------------------------------------------------------------------------------------
template <typename T, bool CONV>
__global__ __launch_bounds__(512) void gemm_tn_pipe4_kernel(const T* __restrict__ dY, const T* __restrict__ X, int M, int N, int K, int ldy,
                                                            int ldx, ConvGeom g, int tiles_k, int tiles, int splits, int m_per_split,
                                                            unsigned ybytes, unsigned xbytes, float* __restrict__ dbias,
                                                            float* __restrict__ slabs, unsigned slab_bytes) {
  static_assert(sizeof(T) == 2, "bf16 only");
  using F = TnFrag<T>;
  constexpr int MSTEP = 64, STAGES = 4, STAGE_B = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);
  const int units = tiles * splits;
  if (first >= units) return;
  const int my_units = (units - first + G - 1) / G;
  const int nsteps = m_per_split / MSTEP;
  const int total = my_units * nsteps;

  if (wave >= 4) {
    // ================================================================ loader waves: pieces {lw, lw + 4, lw + 8, lw + 12} of both operands
    const int lw = wave - 4;
    const rsrc_words_t yr = make_rsrc_words(dY, ybytes), xr = make_rsrc_words(X, xbytes);
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const int lrow = lane >> 4, lpc = lane & 15;
    const int prow0 = lw * 4 + lrow;                              // row (mod 16) of this lane's pieces
    const int fcol = ((((lpc >> 1) ^ tnd_swz(prow0)) << 1) | (lpc & 1)) * 8;   // logical feature column stored at physical chunk lpc
    int is_step = 0, is_unit = first, is_slot = 0;
    bool is_live = true, is_yin = false, is_xin = false;
    unsigned yoff = 0, xoff = 0;
    const unsigned ystride = (unsigned)(MSTEP * ldy * 2), xstride = (unsigned)(MSTEP * ldx * 2);
    const unsigned y16 = (unsigned)(16 * ldy * 2), x16 = (unsigned)(16 * ldx * 2);
    int cv_dy = 0, cv_dx = 0, cv_m = 0;
    long cv_delta = 0;
    const int cv_hw = g.H * g.W;
    const float cv_rhw = 1.f / (float)(CONV ? cv_hw : 1), cv_rw = 1.f / (float)(CONV ? g.W : 1);
    auto enter_unit = [&]() __attribute__((always_inline)) {
      const int tile = is_unit % tiles, split = is_unit / tiles;
      const int tk = tile % tiles_k, tn = tile / tiles_k;
      const int n0 = tn * 128, k0 = tk * 128;
      const long m0 = (long)split * m_per_split + prow0;
      yoff = (unsigned)((m0 * ldy + n0 + fcol) * 2);
      is_yin = n0 + fcol < N; is_xin = k0 + fcol < K;
      if constexpr (!CONV) {
        xoff = (unsigned)((m0 * ldx + k0 + fcol) * 2);
      } else {
        const int kk = k0 + fcol;
        const int tap = kk / g.Ci, ci = kk - tap * g.Ci;
        cv_dy = tap / 3 - 1; cv_dx = tap - (tap / 3) * 3 - 1;
        cv_delta = ((long)(cv_dy * g.W + cv_dx) * g.Ci + ci) * 2;
        cv_m = (int)m0;
      }
    };
    enter_unit();
    auto issue_group = [&]() __attribute__((always_inline)) {   // rows past M fall outside the descriptors: hardware zero fill
      const unsigned st = smem_base + is_slot * STAGE_B + lw * 1024;
      const bool yl = is_live && is_yin, xl = is_live && is_xin;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dma16_asm(yr, st + j * 4096, yl ? yoff + j * y16 : OOB);
        unsigned xo;
        if constexpr (!CONV) {
          xo = xl ? xoff + j * x16 : OOB;
        } else {
          const int m = cv_m + 16 * j;
          const int bimg = div_small(m, cv_hw, cv_rhw);
          const int pix = m - bimg * cv_hw;
          const int y = div_small(pix, g.W, cv_rw), x = pix - y * g.W;
          const bool ok = xl && m < M && (unsigned)(y + cv_dy) < (unsigned)g.H && (unsigned)(x + cv_dx) < (unsigned)g.W;
          xo = ok ? (unsigned)((long)m * g.Ci * 2 + cv_delta) : OOB;
        }
        dma16_asm(xr, st + 16384 + j * 4096, xo);
      }
      is_slot = is_slot + 1 == STAGES ? 0 : is_slot + 1;
      const bool wrap = is_step + 1 == nsteps;
      yoff += ystride; xoff += xstride; cv_m += MSTEP;
      is_step = wrap ? 0 : is_step + 1;
      is_unit = wrap ? is_unit + G : is_unit;
      if (wrap) { if (is_unit < units) enter_unit(); else is_live = false; }
    };
#pragma unroll 1
    for (int i = 0; i < STAGES; ++i) issue_group();
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // group 0
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int k = 0; k < total; ++k) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // group k+1 (groups k+2, k+3 stay in flight)
      __builtin_amdgcn_s_barrier();                         // B_k
      issue_group();                                        // group k+4 -> the slot of stage k
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing no-op pieces
    return;
  }

  // ================================================================== multiplying waves
  const int wn = wave & 1, wk = wave >> 1;
  const __amdgpu_buffer_rsrc_t sr = make_rsrc(slabs, slab_bytes);
  int oa[2][4][2], ob[2][4][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      F::offsets(s2, wn * 64 + i * 16, lane, oa[s2][i][0], oa[s2][i][1]);
      F::offsets(s2, wk * 64 + i * 16, lane, ob[s2][i][0], ob[s2][i][1]);
    }
  struct Frags { s16x4_t a[2][4][2], b[2][4][2]; };   // fragments as their two transpose-read halves: one read per MFMA slot
  Frags fa, fb;
  auto read_half = [&](Frags& f, const char* st, int t) __attribute__((always_inline)) {   // t-th of the 32 reads of a step
    const int idx = t >> 1, h = t & 1, sx = idx >> 3, r = idx & 7;
    if (r < 4) f.b[sx][r][h] = F::load_half(st + 16384, ob[sx][r][h]);
    else f.a[sx][r - 4][h] = F::load_half(st, oa[sx][r - 4][h]);
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum0 = 0.f, bsum1 = 0.f;
  const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};

  __builtin_amdgcn_s_barrier();                             // group 0 has landed
#pragma unroll
  for (int i = 0; i < 32; ++i) read_half(fa, smem, i);

  int st_i = 0, unit = first, rd_slot = 1;
  int c_tk, c_tn, c_split;
  auto locate = [&](int u, int& tn, int& tk, int& split) __attribute__((always_inline)) {
    const int tile = u % tiles;
    split = u / tiles; tn = tile / tiles_k; tk = tile - tn * tiles_k;
  };
  locate(unit, c_tn, c_tk, c_split);
  const int r15 = lane & 15, q = lane >> 4;
  auto step = [&](Frags& cur, Frags& nxt) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this step's fragments (and with them: stage k has been read)
    __builtin_amdgcn_s_barrier();                           // B_k
    __builtin_amdgcn_sched_barrier(0);
    const bool last = __builtin_amdgcn_readfirstlane(st_i + 1) == nsteps;
    const bool bias_unit = dbias != nullptr && c_tk == 0;
    const char* rst = smem + rd_slot * STAGE_B;
    int n_tn = c_tn, n_tk = c_tk, n_split = c_split;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int ms = i >> 4, r = i & 15, ni = r >> 2, ki = r & 3;
      const typename F::Frag fb_ = F::join(cur.b[ms][ki][0], cur.b[ms][ki][1]), fa_ = F::join(cur.a[ms][ni][0], cur.a[ms][ni][1]);
      acc[ni][ki] = Mma<T>::mma(fb_, fa_, acc[ni][ki]);       // D[k][n]: 4 consecutive k per lane
      __builtin_amdgcn_sched_barrier(0);
      read_half(nxt, rst, i);   // (unconditional: past the last step this reads a stale stage into registers nobody uses)
      if (i == 18 && last) locate(unit + G, n_tn, n_tk, n_split);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bias_unit) {   // column sums of dY over this step's 64 rows: this wave's two 16-column blocks of its n half
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        const typename F::Frag v0 = wk == 0 ? F::join(cur.a[ms][0][0], cur.a[ms][0][1]) : F::join(cur.a[ms][2][0], cur.a[ms][2][1]);
        const typename F::Frag v1 = wk == 0 ? F::join(cur.a[ms][1][0], cur.a[ms][1][1]) : F::join(cur.a[ms][3][0], cur.a[ms][3][1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2_t p0 = {v0[2 * e], v0[2 * e + 1]}, p1 = {v1[2 * e], v1[2 * e + 1]};
          bsum0 = __builtin_amdgcn_fdot2_f32_bf16(p0, ones2, bsum0, false);
          bsum1 = __builtin_amdgcn_fdot2_f32_bf16(p1, ones2, bsum1, false);
        }
      }
    }
    if (last) {
      const int n0 = c_tn * 128 + wn * 64, k0 = c_tk * 128 + wk * 64;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
          const int n = n0 + ni * 16 + r15, k = k0 + ki * 16 + q * 4;
          const unsigned o = (n < N && k < K) ? (unsigned)((((long)c_split * N + n) * K + k) * 4) : OOB;   // K % 4 == 0
          bstore16(sr, o, u32x4{__float_as_uint(acc[ni][ki][0]), __float_as_uint(acc[ni][ki][1]), __float_as_uint(acc[ni][ki][2]),
                                __float_as_uint(acc[ni][ki][3])});
          acc[ni][ki] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (bias_unit) {   // fold the 4 m groups of a column (lanes l, l+16, l+32, l+48), then one atomic per column
        float b0 = bsum0, b1 = bsum1;
        b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
        b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
        const int n = n0 + wk * 32 + r15;
        if (q == 0 && n < N) atomicAdd(dbias + n, b0);
        if (q == 0 && n + 16 < N) atomicAdd(dbias + n + 16, b1);
      }
      bsum0 = 0.f; bsum1 = 0.f;
    }
    st_i = last ? 0 : st_i + 1;
    unit = last ? unit + G : unit;
    c_tn = n_tn; c_tk = n_tk; c_split = n_split;
    rd_slot = rd_slot + 1 == STAGES ? 0 : rd_slot + 1;
  };
  for (int gc = 0; gc < total; gc += 2) {
    step(fa, fb);
    if (gc + 1 < total) step(fb, fa);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn_group_kernel (bf16, dense): up to 8 weight-gradient problems with the same row count M in ONE launch, balanced over the CUs.
// All 128 x 128 tiles of all problems form one list (T tiles, S = ceil(M / 64) steps each).  Persistent workgroup c (logical index
// after the XCD remap) first multiplies W = floor(T / G) WHOLE tiles (c W .. c W + W - 1) and adds each straight into dW (load + add +
// store: one owner, no race), then an equal contiguous share of the steps of the remaining R = T - W G tiles; those partial sums go
// to the workgroup's two slab slots (slot 0: the tile was begun by an earlier workgroup, slot 1: it is finished by a later one)
// and tn_group_reduce_kernel folds them into dW.  Why: a stage-3 trunk block has four wgrads of 25-90 tiles each; launched one by
// one they fill 70-88 % of the CUs and each pays a launch ramp and a reduce over 2-3 full-size slabs.  Grouped, every CU gets the
// same number of steps and only the R remainder tiles (19 of 275 for that block) pass through slabs.
// Inner loop = gemm_tn_pipe_kernel's (LDS-DMA through inline asm, transpose-read fragments double-buffered in registers, pinned
// MFMA / read / DMA interleave, dot2 bias sums).
// ------------------------------------------------------------------------------------------------
constexpr int TN_GROUP_MAX = 8;
constexpr int TN_SLOT_FLOATS = 128 * 128;
struct TnJob {
  const void* dY; const void* X; float* dW; float* dbias;
  int N, K, ldy, ldx, ldw;
  int tiles_k, tiles, tile0;  // tile0 = index of this problem's first tile in the group's tile list
};
struct TnGroup {
  TnJob job[TN_GROUP_MAX];
  int njobs, M, S, T;         // S = 64-row steps per tile, T = tiles in all
  int W, RS;                  // whole tiles per workgroup; steps of the remainder tiles (shared evenly)
  int parts;                  // > 0: G = T x parts workgroups, a run = S / parts steps (aligned cuts); physical workgroup ids are part-major
};

__device__ __forceinline__ void tn_locate_tile(const TnGroup& g, int gt, int& j, int& tile) {
  j = 0;
#pragma unroll 1
  for (int i = 1; i < g.njobs; ++i) if (gt >= g.job[i].tile0) j = i;
  tile = gt - g.job[j].tile0;
}

#ifdef SPG_DEV_KERNELS
// In-kernel stamps (DBG 5, tools/tn_stamps.py): per wave, cycles summed over the steps in [top waits | barrier | MFMA block + read drain | tail]
__device__ unsigned long long tn_stamp_sums[256 * 8 * 4];
#endif
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#ifdef SPG_DEV_KERNELS
#include "dev/gemm_tn_group8.inc"
#endif  // SPG_DEV_KERNELS (8-wave grouped wgrad kernel)

// ------------------------------------------------------------------------------------------------
// gemm_tn_group4_kernel: the same grouped schedule (whole tiles, then an equal share of the remainder tiles' steps; slab slots and
// tn_group_reduce as above) with SPECIALISED waves: waves 0-3 multiply (a 64 x 64 quarter of the tile each, one per SIMD), waves 4-7
// only issue the LDS-DMA fill (8 pieces of 1 KiB each per 64-row step).
//
// Why: in-kernel stamps of the 8-wave kernel above (tools/tn_stamps.py, profiles/round2_tn_stamps.md) put a 64-row step at ~2300
// cycles of which the MFMA pipe needs 512: every wave issues, in order, 16 MFMAs + 24 transpose reads + 4 DMA pieces (~80 cycles of
// issue each among reads and MFMAs) + the cursor bookkeeping, and the ablations (no MFMA / no reads / no fill) remove their share
// ADDITIVELY -- the step is bound by each wave's own instruction issue, not by the matrix pipe, LDS or the fill.  With the roles split
// a multiplying wave issues 32 MFMAs + 32 transpose reads and nothing else, and the fill's issue cost sits in a wave of its own
// on the same SIMD (different instruction types issue in the same cycle from different waves).
// Hand-off: one workgroup barrier per step, B_k.  Before it a loader has waited for ITS pieces of group k+1 (vmcnt(16): the two
// younger groups stay in flight), a multiplier for its fragment reads of stage k (lgkmcnt(0)); after it group k+4 may overwrite
// stage k's slot and the fragments of step k+1 may be read.  Both roles pass exactly 1 + total barriers.
// ------------------------------------------------------------------------------------------------
template <typename T, int DBG = 0, int CW = 4>   // CW multiplying waves (4: 64 x 64 each; 8: 64 x 32 each, two per SIMD) + 4 loader waves
__global__ __launch_bounds__((CW + 4) * 64) void gemm_tn_group4_kernel(TnGroup g, float* __restrict__ slabs, unsigned slab_bytes) {
  static_assert(sizeof(T) == 2, "bf16 only");
  using F = TnFrag<T>;
  constexpr int MSTEP = 64, STAGES = 4, STAGE_B = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x;
  // aligned cuts: physical ids (contiguous per XCD) run part-major -- an XCD holds the SAME part of ~32 different tiles, which start at the
  // same row of M, walk in lockstep and share their dY / X panels through that L2; logical c (tile-major) names the slab slots
  const int ci = xcd_remap(blockIdx.x, G);
  const int c = g.parts > 0 ? (ci % g.T) * g.parts + ci / g.T : ci;
  const int S = g.S;
  const int WS = g.W * S;                                                   // steps of this workgroup's whole tiles
  const int rb = (int)((long)c * g.RS / G), re = (int)((long)(c + 1) * g.RS / G);   // its share of the remainder steps
  const int total = WS + (re - rb);
  if (total <= 0) return;
  const int gt_first = WS > 0 ? c * g.W : g.W * G + rb / S;                 // first tile / step of the sequence
  const int m_first = WS > 0 ? 0 : rb % S;
  const int gt_rem = g.W * G + rb / S, m_rem = rb % S;                      // where the remainder share starts

  constexpr int KB = 16 / CW;                                             // 16-column k blocks per multiplying wave (its tile: 64 n x 16 KB k)
  if (wave >= CW) {
    // ================================================================ loader waves
    const int lw = wave - CW;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const int lrow = lane >> 4, lpc = lane & 15;
    const int prow0 = lw * 4 + lrow;                                        // row (mod 16) of this lane's pieces
    const int fcol = ((((lpc >> 1) ^ tnd_swz(prow0)) << 1) | (lpc & 1)) * 8;
    int is_job = -1, is_gt = gt_first, is_tile, is_mstep = m_first, is_ls = 0, is_slot = 0;
    bool is_live = true, is_yin = false, is_xin = false;
    unsigned yoff = 0, xoff = 0, ystride = 0, xstride = 0, y16 = 0, x16 = 0;
    rsrc_words_t yr, xr;
    auto enter_tile = [&]() __attribute__((always_inline)) {   // (also switches descriptors when the tile belongs to another problem)
      int j;
      tn_locate_tile(g, is_gt, j, is_tile);
      const TnJob& jb = g.job[j];
      if (j != is_job) {
        is_job = j;
        yr = make_rsrc_words(jb.dY, (unsigned)((long)g.M * jb.ldy * 2));
        xr = make_rsrc_words(jb.X, (unsigned)((long)g.M * jb.ldx * 2));
        ystride = (unsigned)(MSTEP * jb.ldy * 2); xstride = (unsigned)(MSTEP * jb.ldx * 2);
        y16 = (unsigned)(16 * jb.ldy * 2); x16 = (unsigned)(16 * jb.ldx * 2);
      }
      const int tk = is_tile % jb.tiles_k, tn = is_tile / jb.tiles_k;
      const int n0 = tn * 128, k0 = tk * 128;
      const long m0 = (long)is_mstep * MSTEP + prow0;
      yoff = (unsigned)((m0 * jb.ldy + n0 + fcol) * 2);
      xoff = (unsigned)((m0 * jb.ldx + k0 + fcol) * 2);
      is_yin = n0 + fcol < jb.N; is_xin = k0 + fcol < jb.K;
    };
    enter_tile();
    auto issue_group = [&]() __attribute__((always_inline)) {   // rows past M fall outside the descriptors: hardware zero fill
      const unsigned st = smem_base + is_slot * STAGE_B + lw * 1024;
      const bool yl = is_live && is_yin, xl = is_live && is_xin;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dma16_asm(yr, st + j * 4096, yl ? yoff + j * y16 : OOB);
        dma16_asm(xr, st + 16384 + j * 4096, xl ? xoff + j * x16 : OOB);
      }
      is_slot = is_slot + 1 == STAGES ? 0 : is_slot + 1;
      yoff += ystride; xoff += xstride;
      ++is_ls;
      const bool tile_end = is_mstep + 1 == S;
      if (is_ls >= total) is_live = false;
      else if (is_ls == WS) { is_gt = gt_rem; is_mstep = m_rem; enter_tile(); }        // whole tiles done: jump to the remainder share
      else if (tile_end) { is_gt = is_gt + 1; is_mstep = 0; enter_tile(); }
      else is_mstep = is_mstep + 1;
    };
    for (int i = 0; i < STAGES; ++i) issue_group();
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // group 0
    __builtin_amdgcn_s_barrier();
    unsigned long long sg0 = 0, sg1 = 0, sg2 = 0, tlast = 0;
    if constexpr (DBG == 5) tlast = stamp_now();
    for (int k = 0; k < total; ++k) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // group k+1 (groups k+2, k+3 stay in flight)
      if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg0 += t - tlast; tlast = t; }
      __builtin_amdgcn_s_barrier();                         // B_k
      if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg1 += t - tlast; tlast = t; }
      issue_group();                                        // group k+4 -> the slot of stage k
      if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg2 += t - tlast; tlast = t; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing no-op pieces
#ifdef SPG_DEV_KERNELS
    if constexpr (DBG == 5) {
      if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = tn_stamp_sums + ((int)blockIdx.x * 8 + (wave & 7)) * 4;
        o[0] = sg0; o[1] = sg1; o[2] = sg2; o[3] = 0;
      }
    }
#endif
    return;
  }

  // ================================================================== multiplying waves
  const int wn = wave & 1, wk = wave >> 1;
  const __amdgpu_buffer_rsrc_t sr = make_rsrc(slabs, slab_bytes);
  int oa[2][4][2], ob[2][KB][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) F::offsets(s2, wn * 64 + i * 16, lane, oa[s2][i][0], oa[s2][i][1]);
#pragma unroll
    for (int i = 0; i < KB; ++i) F::offsets(s2, wk * (16 * KB) + i * 16, lane, ob[s2][i][0], ob[s2][i][1]);
  }
  // fragments are kept as their two transpose-read halves: ONE read per MFMA slot fits the ~8 issue cycles an MFMA leaves free
  // (two per slot added their issue time to the MFMA's: stamps, tools/tn_stamps.py)
  struct Frags { s16x4_t a[2][4][2], b[2][KB][2]; };
  Frags fa, fb;
  constexpr int NF = 4 + KB, NRD = 4 * NF, NM = 8 * KB;   // fragments per 32-row half, transpose reads and MFMAs per step
  auto read_half = [&](Frags& f, const char* st, int t) __attribute__((always_inline)) {   // t-th of the NRD reads of a step
    const int idx = t >> 1, h = t & 1, sx = idx / NF, r = idx % NF;
    if (r < KB) f.b[sx][r][h] = F::load_half(st + 16384, ob[sx][r][h]);
    else f.a[sx][r - KB][h] = F::load_half(st, oa[sx][r - KB][h]);
  };
  f32x4 acc[4][KB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum0 = 0.f, bsum1 = 0.f;
  const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};

  __builtin_amdgcn_s_barrier();                             // group 0 has landed
#pragma unroll
  for (int i = 0; i < NRD; ++i) read_half(fa, smem, i);

  int cgt = gt_first, cm = m_first, seg0 = m_first, cls = 0;   // seg0: m step at which this workgroup entered the current tile
  int cj, ctile, c_tk, c_tn;
  bool c_bias;
  auto locate = [&](int gt, int& j, int& tile, int& tn, int& tk, bool& bias) __attribute__((always_inline)) {
    tn_locate_tile(g, gt, j, tile);
    const TnJob& jb = g.job[j];
    tn = tile / jb.tiles_k; tk = tile - tn * jb.tiles_k;
    bias = jb.dbias != nullptr && tk == 0;
  };
  locate(cgt, cj, ctile, c_tn, c_tk, c_bias);
  int rd_slot = 1;
  const int r15 = lane & 15, q = lane >> 4;
  unsigned long long sg0 = 0, sg1 = 0, sg2 = 0, sg3 = 0, tlast = 0;
  if constexpr (DBG == 5) tlast = stamp_now();
  auto step = [&](Frags& cur, Frags& nxt) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this step's fragments (and with them: stage k has been read)
    if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg0 += t - tlast; tlast = t; }
    __builtin_amdgcn_s_barrier();                           // B_k
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg1 += t - tlast; tlast = t; }
    const bool tile_end = __builtin_amdgcn_readfirstlane(cm + 1) == S;
    const bool range_end = cls + 1 == total;
    const bool bias_tile = c_bias;
    const char* rst = smem + rd_slot * STAGE_B;
    // cursor values of the next step, computed in the MFMA shadows below
    int n_cgt = cgt, n_cm = cm, n_seg0 = seg0, n_cj = cj, n_ctile = ctile, n_tn = c_tn, n_tk = c_tk;
    bool n_bias = c_bias;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int ms = i / (4 * KB), r = i % (4 * KB), ni = r / KB, ki = r % KB;
      const typename F::Frag fb_ = F::join(cur.b[ms][ki][0], cur.b[ms][ki][1]), fa_ = F::join(cur.a[ms][ni][0], cur.a[ms][ni][1]);
      if constexpr (DBG != 2) acc[ni][ki] = Mma<T>::mma(fb_, fa_, acc[ni][ki]);
      else asm volatile("" :: "v"(fb_), "v"(fa_));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NRD; ++t)   // (unconditional: past the last step this reads a stale stage into registers nobody uses)
        if (t * NM / NRD == i) read_half(nxt, rst, t);
      if (i == NM / 2 + 2) {
        const int nls = cls + 1;
        const bool jump = nls == WS && nls < total;          // whole tiles done: on to the remainder share
        const bool moved = (jump || tile_end) && !range_end;
        n_cgt = jump ? gt_rem : (tile_end ? cgt + 1 : cgt);
        n_cm = jump ? m_rem : (tile_end ? 0 : cm + 1);
        n_seg0 = jump ? m_rem : (tile_end ? 0 : seg0);
        if (moved) locate(n_cgt, n_cj, n_ctile, n_tn, n_tk, n_bias);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg2 += t - tlast; tlast = t; }
    if (bias_tile) {   // column sums of dY over this step's 64 rows: the 64 / CW... columns of the n half this wave is responsible for
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        typename F::Frag v0, v1;
        if constexpr (CW == 4) {
          v0 = wk == 0 ? F::join(cur.a[ms][0][0], cur.a[ms][0][1]) : F::join(cur.a[ms][2][0], cur.a[ms][2][1]);
          v1 = wk == 0 ? F::join(cur.a[ms][1][0], cur.a[ms][1][1]) : F::join(cur.a[ms][3][0], cur.a[ms][3][1]);
        } else {
          v0 = wk == 0 ? F::join(cur.a[ms][0][0], cur.a[ms][0][1]) : (wk == 1 ? F::join(cur.a[ms][1][0], cur.a[ms][1][1]) :
               (wk == 2 ? F::join(cur.a[ms][2][0], cur.a[ms][2][1]) : F::join(cur.a[ms][3][0], cur.a[ms][3][1])));
          v1 = v0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2_t p0 = {v0[2 * e], v0[2 * e + 1]}, p1 = {v1[2 * e], v1[2 * e + 1]};
          bsum0 = __builtin_amdgcn_fdot2_f32_bf16(p0, ones2, bsum0, false);
          if constexpr (CW == 4) bsum1 = __builtin_amdgcn_fdot2_f32_bf16(p1, ones2, bsum1, false);
        }
      }
    }
    if (tile_end || range_end) {
      const TnJob& jb = g.job[cj];
      const int n0 = c_tn * 128 + wn * 64, k0 = c_tk * 128 + wk * (16 * KB);
      if (seg0 == 0 && tile_end) {
        // the whole tile was multiplied here: accumulate into the gradient (single owner)
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(jb.dW, (unsigned)((long)jb.N * jb.ldw * 4));
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          u32x4 old[KB];
          unsigned o[KB];
#pragma unroll
          for (int ki = 0; ki < KB; ++ki) {
            const int n = n0 + ni * 16 + r15, k = k0 + ki * 16 + q * 4;
            o[ki] = (n < jb.N && k < jb.K) ? (unsigned)(((long)n * jb.ldw + k) * 4) : OOB;   // K % 4 == 0
            old[ki] = bload16(wr, o[ki]);
          }
#pragma unroll
          for (int ki = 0; ki < KB; ++ki) {
            const f32x4 v = acc[ni][ki] + f32x4{__uint_as_float(old[ki].x), __uint_as_float(old[ki].y), __uint_as_float(old[ki].z),
                                                __uint_as_float(old[ki].w)};
            bstore16(wr, o[ki], u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
            acc[ni][ki] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        const int slot = seg0 != 0 ? 0 : 1;
        const unsigned base = (unsigned)(((long)c * 2 + slot) * TN_SLOT_FLOATS * 4);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int ki = 0; ki < KB; ++ki) {
            const int nl = wn * 64 + ni * 16 + r15, kl = wk * (16 * KB) + ki * 16 + q * 4;
            bstore16(sr, base + (unsigned)((nl * 128 + kl) * 4), u32x4{__float_as_uint(acc[ni][ki][0]), __float_as_uint(acc[ni][ki][1]),
                                                                      __float_as_uint(acc[ni][ki][2]), __float_as_uint(acc[ni][ki][3])});
            acc[ni][ki] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      if (bias_tile) {   // partial column sums of this segment: float atomics straight into the bias gradient
        float b0 = bsum0, b1 = bsum1;
        b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
        b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
        const int n = n0 + wk * (CW == 4 ? 32 : 16) + r15;
        if (q == 0 && n < jb.N) atomicAdd(jb.dbias + n, b0);
        if (CW == 4 && q == 0 && n + 16 < jb.N) atomicAdd(jb.dbias + n + 16, b1);
      }
      bsum0 = 0.f; bsum1 = 0.f;
    }
    cgt = n_cgt; cm = n_cm; seg0 = n_seg0; cls = cls + 1;
    cj = n_cj; ctile = n_ctile; c_tn = n_tn; c_tk = n_tk; c_bias = n_bias;
    rd_slot = rd_slot + 1 == STAGES ? 0 : rd_slot + 1;
    if constexpr (DBG == 5) { const unsigned long long t = stamp_now(); sg3 += t - tlast; tlast = t; }
  };
  for (int gc = 0; gc < total; gc += 2) {
    step(fa, fb);
    if (gc + 1 < total) step(fb, fa);
  }
#ifdef SPG_DEV_KERNELS
  if constexpr (DBG == 5) {
    if (lane == 0 && blockIdx.x < 256) {
      unsigned long long* o = tn_stamp_sums + ((int)blockIdx.x * 8 + wave) * 4;
      o[0] = sg0; o[1] = sg1; o[2] = sg2; o[3] = sg3;
    }
  }
#endif
}

// Folds the partial sums of the remainder tiles cut by share boundaries into dW.  Block b looks at boundary c = b + 1 (between the
// remainder shares of logical workgroups c-1 and c); it acts only if that boundary falls inside a tile AND is the first one inside.
__device__ __forceinline__ void tn_group_reduce_body(const TnGroup& g, const float* __restrict__ slabs, int G, int c, int ypart, int yparts) {
  const int S = g.S;
  const int b = (int)((long)c * g.RS / G);
  if (b >= g.RS) return;
  const int mstep = b % S;
  if (mstep == 0) return;                                   // the tile starts exactly at the boundary: not cut here
  const int tile_start = b - mstep;
  if ((int)((long)(c - 1) * g.RS / G) > tile_start) return;      // an earlier boundary already lies inside this tile
  int j, tile;
  tn_locate_tile(g, g.W * G + b / S, j, tile);
  const TnJob& jb = g.job[j];
  const int tile_endg = tile_start + S;
  int clast = c;                                             // last workgroup touching the tile
  while (clast + 1 < G && (int)((long)(clast + 1) * g.RS / G) < tile_endg) ++clast;
  const int tk = tile % jb.tiles_k, tn = tile / jb.tiles_k;
  // gridDim.y blocks share the tile's 4096 float4 (a remainder tile can be cut into a dozen pieces: one block alone would walk them
  // all at a single CU's bandwidth)
  const int per = (TN_SLOT_FLOATS / 4) / yparts;
  for (int v = ypart * per + threadIdx.x; v < (ypart + 1) * per; v += 256) {
    const int nl = v >> 5, kl = (v & 31) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + ((long)(c - 1) * 2 + 1) * TN_SLOT_FLOATS + v * 4);
    // (workgroups whose remainder share is empty -- RS < G, i.e. few rows -- wrote no slot: skipping them matters, their slots hold
    // whatever an earlier launch left in the scratch)
#pragma unroll 4
    for (int cc = c; cc <= clast; ++cc)
      if ((int)((long)(cc + 1) * g.RS / G) > (int)((long)cc * g.RS / G))
        s += *reinterpret_cast<const f32x4*>(slabs + ((long)cc * 2 + 0) * TN_SLOT_FLOATS + v * 4);
    const int n = tn * 128 + nl, k = tk * 128 + kl;
    if (n < jb.N && k < jb.K) {
      float* d = jb.dW + (long)n * jb.ldw + k;
      *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + s;
    }
  }
}

__global__ __launch_bounds__(256) void tn_group_reduce_kernel(TnGroup g, const float* __restrict__ slabs, int G) {
  tn_group_reduce_body(g, slabs, G, blockIdx.x + 1, blockIdx.y, gridDim.y);
}

// the boundary reduces of up to 6 grouped launches in one launch (they only feed the optimizer: the trunk backward defers them)
constexpr int TN_REDUCE_BATCH = 6;
struct TnReduceDesc { TnGroup g; int G; int pad; };
struct TnReduceBatch { TnGroup g[TN_REDUCE_BATCH]; const float* slabs[TN_REDUCE_BATCH]; int G[TN_REDUCE_BATCH]; int n; };
__global__ __launch_bounds__(256) void tn_group_reduce_batch_kernel(TnReduceBatch b) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x + 1 >= b.G[z]) return;
  tn_group_reduce_body(b.g[z], b.slabs[z], b.G[z], blockIdx.x + 1, blockIdx.y, gridDim.y);
}

#ifdef SPG_DEV_KERNELS
#include "dev/gemm_tn_wide.inc"
#endif  // SPG_DEV_KERNELS (wide grouped wgrad kernel)

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_matrix_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C, int transpose) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  if (!transpose) {
    for (int j = ty; j < 32; j += 8) {
      const int r = by + j, c = bx + tx;
      if (r < R && c < C) ST<T>::st(dst + (long)r * C + c, src[(long)r * C + c]);
    }
    return;
  }
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;  // dst[c][r]
    if (r < R && c < C) ST<T>::st(dst + (long)c * R + r, tile[tx][j]);
  }
}

// Batched form of pack_matrix: one launch re-packs every Linear / 1x1 weight of the model (job table in device memory;
// a block finds its job by binary search over the tile prefix sums).  Removes ~400 tiny launches per optimizer step.
struct PackJob {
  const float* src;
  void* dst;      // [R][C] copy (may be null)
  void* dst_t;    // [C][R] transposed copy (may be null)
  int R, C, tile0, pad;   // tile0 = first 32x32 tile index of this job
};
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {  // last job with tile0 <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const PackJob jb = jobs[lo];
  const int t = b - jb.tile0;
  const int tiles_c = (jb.C + 31) >> 5;
  const int bx = (t % tiles_c) * 32, by = (t / tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* dst = reinterpret_cast<T*>(jb.dst);
  T* dst_t = reinterpret_cast<T*>(jb.dst_t);
  const int R = jb.R, C = jb.C;
  for (int j = ty; j < 32; j += 8) {          // one read of the fp32 master serves both layouts
    const int r = by + j, c = bx + tx;
    const float v = (r < R && c < C) ? jb.src[(long)r * C + c] : 0.f;
    tile[j][tx] = v;
    if (dst && r < R && c < C) ST<T>::st(dst + (long)r * C + c, v);
  }
  if (!dst_t) return;
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (r < R && c < C) ST<T>::st(dst_t + (long)c * R + r, tile[tx][j]);
  }
}

template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ src, T* __restrict__ fwd, T* __restrict__ dgr, int Co,
                                    int Ci) {
  const long n = (long)Co * Ci * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long t = i / 9;
    const int ci = (int)(t % Ci), co = (int)(t / Ci);
    const float v = src[i];
    if (fwd) ST<T>::st(fwd + ((long)co * 9 + tap) * Ci + ci, v);
    if (dgr) ST<T>::st(dgr + ((long)ci * 9 + (8 - tap)) * Co + co, v);
  }
}

__global__ void unpack_conv3x3_grad_kernel(const float* __restrict__ packed, float* __restrict__ dst, int Co, int Ci) {
  const long n = (long)Co * Ci * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long t = i / 9;
    const int ci = (int)(t % Ci), co = (int)(t / Ci);
    dst[i] += packed[((long)co * 9 + tap) * Ci + ci];
  }
}

// CUs the persistent GEMM grids are sized for.  These kernels occupy a CU completely (160 KiB LDS, every VGPR), so when another
// long-running kernel holds some CUs -- RCCL's all-reduce while it overlaps the backward pass -- a grid sized to all 256 needs a second
// round for the displaced workgroups (2x for every GEMM that overlaps the collective).  Every GEMM entry point therefore takes a
// cu_budget argument (0 = all CUs): per call, no process-global state (the multi-GPU trainer passes it for the graph segments that
// run beside a collective, engine/trainer.py).
// 1 = product rule (v3 where it measured faster), dev builds: SPG_NT_V3=0 never, 2 always (A/B runs)
static inline int nt_v3_mode() {
#ifdef SPG_DEV_KERNELS
  static const int v = dev_env("SPG_NT_V3", 1);
  return v;
#else
  return 1;
#endif
}
static inline bool nt_v3_enabled() { return nt_v3_mode() != 0; }
#ifdef SPG_DEV_KERNELS
static inline int nt_v5_mode() {   // SPG_NT_V5: 1 routes every applicable NT problem to gemm_nt_v5, 2 only the long-K ones, 0 none
  static const int v = dev_env("SPG_NT_V5", 0);
  return v;
}
#endif
static inline bool tn_group_v4() {   // dev builds: SPG_TN_GROUP_V4=0 selects the 8-wave kernel (A/B runs)
#ifdef SPG_DEV_KERNELS
  static const int v = dev_env("SPG_TN_GROUP_V4", 1);
  return v != 0;
#else
  return true;
#endif
}
static int hw_cus() {
  static int hw = 0;
  if (hw == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) hw = p.multiProcessorCount;
    if (hw <= 0) hw = 256;
#ifdef SPG_DEV_KERNELS
    const int v = dev_env("SPG_CUS", 0);
    if (v >= 8 && v < hw) hw = v;
#endif
  }
  return hw;
}
static inline int num_cus(int budget = 0) {
  const int hw = hw_cus();
  return (budget >= 8 && budget < hw) ? budget : hw;
}
// tile width (32 * nb columns) of the persistent NT kernels: minimise rounds(tiles / CUs) x per-tile cost (MFMA work ~ nb, X fill + fixed ~ 1.5)
static inline int pick_nb(int N, int tiles_m, int cus) {
  int nb = 4;
  float best = 1e30f;
  for (int c = 4; c >= 2; --c) {
    const long t = (long)cdiv(N, 32 * c) * tiles_m;
    const float cost = (float)cdiv(t, cus) * ((float)c + 1.5f);
    if (cost < best * 0.999f) { best = cost; nb = c; }
  }
  return nb;
}


// bf16 3x3 convolutions without a fused activation / residual go to the halo-tile kernel (conv_halo.hip) when it has an instance
// for them; 1 = not taken.  Dev builds: SPG_CONV_HALO=0 keeps them on the implicit GEMM, 64 / 128 / 256 force that tile width.
static int try_conv_halo(const void* X, const void* W, void* C, const NtEpi& epi, int N, int ldc, int conv, const ConvGeom& g, int cus,
                         hipStream_t s) {
  if (!conv || epi.act != SPG_ACT_NONE || epi.residual || epi.gelu_h || epi.C2) return 1;
  int force = 0;
#ifdef SPG_DEV_KERNELS
  static const int mode = dev_env("SPG_CONV_HALO", 1);
  if (mode == 0) return 1;
  static const int dbg = dev_env("SPG_CONV_HALO_DBG", 0);   // ablation instances of conv_halo.hip (1 no DMA, 2 no reads, 3 no MFMAs, 4 no stores)
  force = (mode == 1 ? 0 : mode) + 1000 * dbg;
#endif
  return launch_conv3x3_halo(X, W, C, epi.bias, g.B, g.H, g.W, g.Ci, N, ldc, cus, force, s, nullptr);
}

// Dev builds, SPG_NT_WIDE=1: dense bf16 problems with a wide output and no residual go to the 192-column variable-height kernel
// (nt_wide.hip) when its plan fills the CUs; 1 = not taken.  A measured experiment (no gain in the step), not a product path.
static int try_nt_wide(const void* X, const void* W, void* C, const NtEpi& epi, int M, int N, int K, int ldx, int ldc, int conv, int cus,
                       hipStream_t s) {
#ifdef SPG_DEV_KERNELS
  static const int mode = dev_env("SPG_NT_WIDE", 0);
  if (mode == 0) return 1;
  if (conv || epi.residual || epi.act == SPG_ACT_RELU) return 1;
  int pact;
  if (epi.gelu_h) {
    if (epi.act != SPG_ACT_MUL_H) return 1;
    pact = 4;   // PIPE_ACT_MULH
  } else {
    pact = epi.act == SPG_ACT_GELU ? 1 : (epi.act == SPG_ACT_GELU_SAVE_GRAD ? 3 : (epi.act == SPG_ACT_NONE ? 0 : -1));
    if (pact < 0) return 1;
  }
  static const int dbg = dev_env("SPG_NT_WIDE_DBG", 0);   // 5: in-kernel stamps (tools/ntw_stamps.py)
  return launch_nt_wide(X, W, C, epi.bias, epi.gelu_h, epi.C2, pact, M, N, K, ldx, ldc, cus, dbg, s);
#else
  (void)X; (void)W; (void)C; (void)epi; (void)M; (void)N; (void)K; (void)ldx; (void)ldc; (void)conv; (void)cus; (void)s;
  return 1;
#endif
}

// bf16 problems the 4-wave two-per-CU kernel has an instance for: 8-element-aligned rows, operands addressable by 32-bit offsets, no ReLU
// (NT_V3_NA = not applicable, the caller falls through to the older kernels)
constexpr int NT_V3_NA = -1000;
#ifndef SPG_V3_KS   // (tools/ builds may override the stage geometry: -DSPG_V3_KS=1 -DSPG_V3_NS=4)
#define SPG_V3_KS 2
#define SPG_V3_NS 2
#endif
constexpr int V3_KS = SPG_V3_KS, V3_NS = SPG_V3_NS;      // 2 stages of 32 KiB (NB = 4): 64 KiB per workgroup, two workgroups per CU
static int launch_nt_v3(const void* X, const void* W, void* C, NtEpi epi, int M, int N, int K, int ldx, int ldc, int conv, ConvGeom g,
                        hipStream_t s, int cu_budget) {
  const int tiles_m = cdiv(M, BM);
  const long xb = (conv ? (long)M * g.Ci : (long)M * ldx) * 2L, wb = (long)N * K * 2L;
  const long cb = ((long)(M - 1) * ldc + N) * 2;
  const int pact = epi.gelu_h ? (epi.act == SPG_ACT_MUL_H ? PIPE_ACT_MULH : PIPE_ACT_HH)
                                 : (epi.act == SPG_ACT_GELU ? PIPE_ACT_GELU : (epi.act == SPG_ACT_GELU_SAVE_GRAD ? PIPE_ACT_GELU_D : PIPE_ACT_NONE));
  const bool epi_ok = epi.act != SPG_ACT_RELU && !(epi.gelu_h && epi.act != SPG_ACT_NONE && epi.act != SPG_ACT_MUL_H) &&
                      (epi.C2 == nullptr || pact == PIPE_ACT_GELU || pact == PIPE_ACT_GELU_D) && (pact != PIPE_ACT_GELU_D || epi.C2 != nullptr) &&
                      !(conv && pact != PIPE_ACT_NONE);
  if (nt_v3_enabled() && N % 8 == 0 && ldc % 8 == 0 && K % 8 == 0 && (conv ? g.Ci % 8 == 0 : ldx % 8 == 0) && cb < 0xFFFFFFF0L &&
      xb < 0xFFFFFFF0L && wb < 0xFFFFFFF0L && epi_ok) {
      PipeEpi pe;
      pe.c_bytes = (unsigned)cb;
      pe.c2_bytes = epi.C2 ? (unsigned)cb : 0u;
      pe.r_bytes = epi.residual ? (unsigned)cb : 0u;
      pe.h_bytes = epi.gelu_h ? (unsigned)cb : 0u;
      pe.bias_bytes = epi.bias ? (unsigned)N * 4u : 0u;
      pe.pf = t_cur_pf; pe.pf_bytes = t_cur_pf_bytes;
      // tile width: columns of work (incl. the zero columns of the last tile) x per-column-tile fixed cost (X fill, prologue, epilogue)
#ifdef SPG_V3_FORCE_NB     // (tools/ A/B builds)
      const int nb3 = SPG_V3_FORCE_NB;
#else
      const int nb3 = cdiv(N, 128) * 5 <= cdiv(N, 64) * 3 ? 4 : 2;
#endif
      const int tn3 = cdiv(N, 32 * nb3);
      const int grid3 = tn3 * tiles_m;
      // Measured (tools/nt_check.py, hipGraph timing): with two workgroups on most CUs and a short K loop this kernel beats the
      // persistent one by 5-25 % (qkv 17.8 -> 16.2 us, fc1+GELU 35.4 -> 30.4, the stage-1/2 projections 25-30 %); on grids that leave
      // one 4-wave workgroup per CU, or with long K loops where the persistent kernel's three DMA groups in flight pay (fc2, dqkv,
      // the CFI fusion GEMM), it loses 10-60 %.  Dispatch on exactly that.
      const int cus3 = hw_cus();
#ifdef SPG_DEV_KERNELS
      if (pact <= PIPE_ACT_HH && (nt_v5_mode() == 1 || (nt_v5_mode() == 2 && !conv && K > 1536))) {   // A/B runs of the specialised-wave NT kernel
        const int cus5 = num_cus(cu_budget);
        const int grid5 = grid3 < cus5 ? grid3 : cus5;
#define SPG_LAUNCH5(C_, A_, NB_)                                                                                                           \
  do {                                                                                                                                     \
    constexpr int lds_ = 4 * (BM + 32 * NB_) * 128;                                                                                        \
    static bool attr_ = false;                                                                                                             \
    if (!attr_) {                                                                                                                          \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v5_kernel<C_, A_, NB_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_); \
      attr_ = true;                                                                                                                        \
    }                                                                                                                                      \
    hipLaunchKernelGGL((gemm_nt_v5_kernel<C_, A_, NB_>), dim3(grid5), dim3(512), lds_, s, (const bf16_t*)X, (const bf16_t*)W, (bf16_t*)C, \
                       epi, pe, M, N, K, ldx, ldc, g, tn3, grid3, (unsigned)xb, (unsigned)wb);                                             \
  } while (0)
#define SPG_LAUNCH5_NB(C_, A_) \
  do { if (nb3 == 4) SPG_LAUNCH5(C_, A_, 4); else SPG_LAUNCH5(C_, A_, 2); } while (0)
        if (conv) SPG_LAUNCH5_NB(true, PIPE_ACT_NONE);
        else if (pact == PIPE_ACT_GELU) SPG_LAUNCH5_NB(false, PIPE_ACT_GELU);
        else if (pact == PIPE_ACT_HH) SPG_LAUNCH5_NB(false, PIPE_ACT_HH);
        else SPG_LAUNCH5_NB(false, PIPE_ACT_NONE);
#undef SPG_LAUNCH5_NB
#undef SPG_LAUNCH5
        return check_launch("gemm_nt(v5)");
      }
#endif
      if (nt_v3_mode() == 1 && (grid3 < cus3 + cus3 / 4 || K > 1536)) return NT_V3_NA;
#define SPG_LAUNCH3(C_, A_, NB_)                                                                                                           \
  do {                                                                                                                                     \
    constexpr int lds_ = V3_NS * (BM + 32 * NB_) * 64 * V3_KS;                                                                                       \
    static bool attr_ = false;                                                                                                             \
    if (!attr_) {                                                                                                                          \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v3_kernel<C_, A_, NB_, V3_KS, V3_NS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_); \
      attr_ = true;                                                                                                                        \
    }                                                                                                                                      \
    hipLaunchKernelGGL((gemm_nt_v3_kernel<C_, A_, NB_, V3_KS, V3_NS>), dim3(grid3), dim3(256), lds_, s, (const bf16_t*)X, (const bf16_t*)W, (bf16_t*)C, \
                       epi, pe, M, N, K, ldx, ldc, g, tn3, (unsigned)xb, (unsigned)wb);                                                    \
  } while (0)
#define SPG_LAUNCH3_NB(C_, A_) \
  do { if (nb3 == 4) SPG_LAUNCH3(C_, A_, 4); else SPG_LAUNCH3(C_, A_, 2); } while (0)
      if (conv) SPG_LAUNCH3_NB(true, PIPE_ACT_NONE);
      else if (pact == PIPE_ACT_GELU) SPG_LAUNCH3_NB(false, PIPE_ACT_GELU);
      else if (pact == PIPE_ACT_HH) SPG_LAUNCH3_NB(false, PIPE_ACT_HH);
      else if (pact == PIPE_ACT_GELU_D) SPG_LAUNCH3_NB(false, PIPE_ACT_GELU_D);
      else if (pact == PIPE_ACT_MULH) SPG_LAUNCH3_NB(false, PIPE_ACT_MULH);
      else SPG_LAUNCH3_NB(false, PIPE_ACT_NONE);
#undef SPG_LAUNCH3_NB
#undef SPG_LAUNCH3
      return check_launch("gemm_nt(v3)");
  }
  return NT_V3_NA;
}

#ifdef SPG_DEV_KERNELS
#include "dev/launch_nt_dev.inc"
#else
// Product dispatch: the pipelined persistent kernel for bf16 problems it has an instance for, the 8-wave LDS-DMA kernel for everything
// else (fp32 parity mode; bf16 with ReLU, one K step, or rows that are not 8-element aligned).
template <typename T>
static int launch_nt(const void* X, const void* W, void* C, NtEpi epi, int M, int N, int K, int ldx, int ldc, int conv,
                     ConvGeom g, hipStream_t s, int cu_budget) {
  const int tiles_m = cdiv(M, BM);
  const int cus = num_cus(cu_budget);
  const long xb = (conv ? (long)M * g.Ci : (long)M * ldx) * (long)sizeof(T), wb = (long)N * K * (long)sizeof(T);
  if (xb >= 0xFFFFFFF0L || wb >= 0xFFFFFFF0L) {
    set_error("gemm_nt: operand larger than 4 GiB (X %ld B, W %ld B) is not addressable by one buffer descriptor", xb, wb);
    return SPG_ERR_UNSUPPORTED;
  }
  const int nb = pick_nb(N, tiles_m, cus);
  const int tn_ = cdiv(N, 32 * nb);
  const int nwg_ = tn_ * tiles_m;
  const int grid = nwg_ < cus ? nwg_ : cus;
  if constexpr (sizeof(T) == 2) {
    // pipelined kernel: bf16, >= 2 K steps per tile, 8-element-aligned rows, operands addressable by 32-bit offsets, and an
    // epilogue it has an instance for (ReLU, or GELU together with gelu_h, go to the plain DMA kernel below)
    const long cb = ((long)(M - 1) * ldc + N) * 2;
    const int pact = epi.gelu_h ? (epi.act == SPG_ACT_MUL_H ? PIPE_ACT_MULH : PIPE_ACT_HH)
                                 : (epi.act == SPG_ACT_GELU ? PIPE_ACT_GELU : (epi.act == SPG_ACT_GELU_SAVE_GRAD ? PIPE_ACT_GELU_D : PIPE_ACT_NONE));
    const bool epi_ok = epi.act != SPG_ACT_RELU && !(epi.gelu_h && epi.act != SPG_ACT_NONE && epi.act != SPG_ACT_MUL_H) &&
                      (epi.C2 == nullptr || pact == PIPE_ACT_GELU || pact == PIPE_ACT_GELU_D) && (pact != PIPE_ACT_GELU_D || epi.C2 != nullptr) &&
                        !(conv && pact != PIPE_ACT_NONE);
    {
      const int rch = try_conv_halo(X, W, C, epi, N, ldc, conv, g, cus, s);
      if (rch != 1) return rch;
      const int rcw = try_nt_wide(X, W, C, epi, M, N, K, ldx, ldc, conv, cus, s);
      if (rcw != 1) return rcw;
      const int rc3 = launch_nt_v3(X, W, C, epi, M, N, K, ldx, ldc, conv, g, s, cu_budget);
      if (rc3 != NT_V3_NA) return rc3;
    }
    if (K > ROWB / (int)sizeof(T) && N % 8 == 0 && ldc % 8 == 0 && cb < 0xFFFFFFF0L && epi_ok) {
      PipeEpi pe;
      pe.c_bytes = (unsigned)cb;
      pe.c2_bytes = epi.C2 ? (unsigned)cb : 0u;
      pe.r_bytes = epi.residual ? (unsigned)cb : 0u;
      pe.h_bytes = epi.gelu_h ? (unsigned)cb : 0u;
      pe.bias_bytes = epi.bias ? (unsigned)N * 4u : 0u;
      pe.pf = t_cur_pf; pe.pf_bytes = t_cur_pf_bytes;
#define SPG_LAUNCHP(C_, A_, NB_)                                                                                                             \
  do {                                                                                                                                     \
    static bool attr_ = false;                                                                                                             \
    if (!attr_) {                                                                                                                          \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pipe_kernel<T, C_, A_, NB_, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                          PIPE_LDS_BYTES);                                                                                                 \
      attr_ = true;                                                                                                                        \
    }                                                                                                                                      \
    hipLaunchKernelGGL((gemm_nt_pipe_kernel<T, C_, A_, NB_, false, 0>), dim3(grid), dim3(512), PIPE_LDS_BYTES, s, (const T*)X, (const T*)W,  \
                       (T*)C, epi, pe, M, N, K, ldx, ldc, g, tn_, nwg_, (unsigned)xb, (unsigned)wb);                                       \
  } while (0)
#define SPG_LAUNCHP_NB(C_, A_) \
  do { if (nb == 4) SPG_LAUNCHP(C_, A_, 4); else if (nb == 3) SPG_LAUNCHP(C_, A_, 3); else SPG_LAUNCHP(C_, A_, 2); } while (0)
      if (conv) SPG_LAUNCHP_NB(true, PIPE_ACT_NONE);
      else if (pact == PIPE_ACT_GELU) SPG_LAUNCHP_NB(false, PIPE_ACT_GELU);
      else if (pact == PIPE_ACT_HH) SPG_LAUNCHP_NB(false, PIPE_ACT_HH);
      else if (pact == PIPE_ACT_GELU_D) SPG_LAUNCHP_NB(false, PIPE_ACT_GELU_D);
      else if (pact == PIPE_ACT_MULH) SPG_LAUNCHP_NB(false, PIPE_ACT_MULH);
      else SPG_LAUNCHP_NB(false, PIPE_ACT_NONE);
#undef SPG_LAUNCHP_NB
#undef SPG_LAUNCHP
      return check_launch("gemm_nt(pipe)");
    }
  }
  if (epi.act == SPG_ACT_GELU_SAVE_GRAD || epi.act == SPG_ACT_MUL_H) {
    set_error("gemm_nt: act %d (saved GELU derivative) exists only in the bf16 pipelined kernels: needs bf16, K > 64, 8-element-aligned "
              "N / K / ldc / ldx (M=%d N=%d K=%d)", epi.act, M, N, K);
    return SPG_ERR_UNSUPPORTED;
  }
  constexpr int LDS8 = 3 * DMA_STAGE_BYTES + 8 * 16 * 68 * 4;
  static bool attr8 = false;
  if (!attr8) {
#define SPG_SET_ATTR(C_, NB_) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_dma_kernel<T, C_, 0, 4, NB_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8)
    SPG_SET_ATTR(true, 4); SPG_SET_ATTR(false, 4); SPG_SET_ATTR(true, 3); SPG_SET_ATTR(false, 3); SPG_SET_ATTR(true, 2); SPG_SET_ATTR(false, 2);
#undef SPG_SET_ATTR
    attr8 = true;
  }
#define SPG_LAUNCH8(C_, NB_) hipLaunchKernelGGL((gemm_nt_dma_kernel<T, C_, 0, 4, NB_>), dim3(grid), dim3(512), LDS8, s, (const T*)X, (const T*)W, \
                           (T*)C, epi, M, N, K, ldx, ldc, g, tn_, nwg_, (unsigned)xb, (unsigned)wb)
  if (conv) { if (nb == 4) SPG_LAUNCH8(true, 4); else if (nb == 3) SPG_LAUNCH8(true, 3); else SPG_LAUNCH8(true, 2); }
  else { if (nb == 4) SPG_LAUNCH8(false, 4); else if (nb == 3) SPG_LAUNCH8(false, 3); else SPG_LAUNCH8(false, 2); }
#undef SPG_LAUNCH8
  return check_launch("gemm_nt(dma8)");
}
#endif  // SPG_DEV_KERNELS

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int splits, long nk4,
                                                        int K, int ldw) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nk4; i += (long)gridDim.x * 256) {
    // (eight slabs' loads in flight, added in slab order: the sum is the one the one-load-per-trip loop formed, without its `splits`
    // dependent memory round trips -- 19 us for a 19-way split of a 64 k-element gradient)
    f32x4 a = *reinterpret_cast<const f32x4*>(slabs + i * 4);
    int sidx = 1;
    for (; sidx + 8 <= splits; sidx += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(slabs + ((long)(sidx + j) * nk4 + i) * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; sidx < splits; ++sidx) a += *reinterpret_cast<const f32x4*>(slabs + ((long)sidx * nk4 + i) * 4);
    const long e = i * 4;
    float* d = dW + (e / K) * ldw + (e % K);      // K % 4 == 0, so the 4 elements stay inside one row
    f32x4 o = *reinterpret_cast<f32x4*>(d);
    *reinterpret_cast<f32x4*>(d) = o + a;
  }
}

template <typename T>
static void tn_split_plan(int M, int N, int K, int* splits, int* m_per_split) {
  constexpr int MSTEP = ROWB / (int)sizeof(T);
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
#ifdef SPG_DEV_KERNELS
  static const int target = dev_env("SPG_TN_TARGET", 384);
#else
  constexpr int target = 384;
#endif
  int sp = cdiv(target, tiles);  // blocks ~ target
  const int max_splits = cdiv(M, 4 * MSTEP);
  if (sp > max_splits) sp = max_splits;
  if (sp < 1) sp = 1;
  const int mps = cdiv(cdiv(M, sp), MSTEP) * MSTEP;
  *m_per_split = mps;
  *splits = cdiv(M, mps);
}

template <typename T>
static int launch_tn(const void* dY, const void* X, float* dW, int M, int N, int K, int ldy, int ldx, int ldw, int conv,
                     ConvGeom g, hipStream_t s, float* dbias, float* ws, size_t ws_bytes, int cu_budget) {
  constexpr int MSTEP = ROWB / (int)sizeof(T);
  const int tiles_n = cdiv(N, 128), tiles_k = cdiv(K, 128);
  const int tiles = tiles_n * tiles_k;
  const int max_splits = cdiv(M, 4 * MSTEP);
  int splits, m_per_split;
  tn_split_plan<T>(M, N, K, &splits, &m_per_split);
  float* slabs = nullptr;
  if (ws && splits > 1 && ws_bytes >= (size_t)splits * N * K * sizeof(float)) slabs = ws;
  const size_t lds = 4 * 128 * ROWB;
  const long yb = (long)M * ldy * (long)sizeof(T), xb = (conv ? (long)M * g.Ci : (long)M * ldx) * (long)sizeof(T);
  if (xb >= 0xFFFFFFF0L || yb >= 0xFFFFFFF0L) {
    set_error("gemm_tn: operand larger than 4 GiB is not addressable by one buffer descriptor");
    return SPG_ERR_UNSUPPORTED;
  }
#ifdef SPG_DEV_KERNELS
  static const int tnv = dev_env("SPG_GEMM_TN_STAGED", 0) ? 0 : 2;   // A/B runs: the register-staged kernel for bf16 too
#else
  constexpr int tnv = 2;                                             // pipelined kernel wherever it applies
#endif
  if constexpr (sizeof(T) == 2) {
    if (tnv == 2 && K % 4 == 0 && ws && (!conv || M < (1 << 24))) {
      int sp = num_cus(cu_budget) / tiles;             // one unit per workgroup is the balanced case
      if (sp > max_splits) sp = max_splits;
      if (sp < 1) sp = 1;
      const int mps = cdiv(cdiv(M, sp), MSTEP) * MSTEP;
      sp = cdiv(M, mps);
      const size_t need = (size_t)sp * N * K * sizeof(float);
      if (ws_bytes >= need && need < 0xFFFFFFF0UL) {
        const int units = tiles * sp;
        const int grid = units < num_cus(cu_budget) ? units : num_cus(cu_budget);
        constexpr int LDSP = 4 * 32768;
#ifdef SPG_DEV_KERNELS
        static bool attrp = false;
        if (!attrp) {
          hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pipe_kernel<T, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
          hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pipe_kernel<T, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
          attrp = true;
        }
#endif
        if (tn_group_v4()) {   // specialised waves (dev builds: SPG_TN_GROUP_V4=0 selects the 8-wave kernels)
          static bool attr4 = false;
          if (!attr4) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pipe4_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pipe4_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
            attr4 = true;
          }
          if (conv)
            hipLaunchKernelGGL((gemm_tn_pipe4_kernel<T, true>), dim3(grid), dim3(512), LDSP, s, (const T*)dY, (const T*)X, M, N, K, ldy, ldx, g,
                               tiles_k, tiles, sp, mps, (unsigned)yb, (unsigned)xb, dbias, ws, (unsigned)need);
          else
            hipLaunchKernelGGL((gemm_tn_pipe4_kernel<T, false>), dim3(grid), dim3(512), LDSP, s, (const T*)dY, (const T*)X, M, N, K, ldy, ldx, g,
                               tiles_k, tiles, sp, mps, (unsigned)yb, (unsigned)xb, dbias, ws, (unsigned)need);
        }
#ifdef SPG_DEV_KERNELS
        else if (conv)
          hipLaunchKernelGGL((gemm_tn_pipe_kernel<T, true, 0>), dim3(grid), dim3(512), LDSP, s, (const T*)dY, (const T*)X, M, N, K, ldy, ldx, g,
                             tiles_k, tiles, sp, mps, (unsigned)yb, (unsigned)xb, dbias, ws, (unsigned)need);
        else
          hipLaunchKernelGGL((gemm_tn_pipe_kernel<T, false, 0>), dim3(grid), dim3(512), LDSP, s, (const T*)dY, (const T*)X, M, N, K, ldy, ldx, g,
                             tiles_k, tiles, sp, mps, (unsigned)yb, (unsigned)xb, dbias, ws, (unsigned)need);
#endif
        int rc = check_launch("gemm_tn(pipe)");
        if (rc) return rc;
        const long nk4 = (long)N * K / 4;
        long gr = (nk4 + 255) / 256;
        if (gr > 2048) gr = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((int)gr), dim3(256), 0, s, ws, dW, sp, nk4, K, ldw);
        return check_launch("gemm_tn(pipe reduce)");
      }
    }
  }
#ifdef SPG_DEV_KERNELS
  if (dev_env("SPG_TN_WAVES", 4) == 8) {
    if (conv)
      hipLaunchKernelGGL((gemm_tn_kernel<T, true, 4>), dim3(tiles, splits), dim3(512), lds, s, (const T*)dY,
                         (const T*)X, dW, M, N, K, ldy, ldx, ldw, g, tiles_k, m_per_split, (unsigned)yb, (unsigned)xb, dbias, nullptr);
    else
      hipLaunchKernelGGL((gemm_tn_kernel<T, false, 4>), dim3(tiles, splits), dim3(512), lds, s, (const T*)dY,
                         (const T*)X, dW, M, N, K, ldy, ldx, ldw, g, tiles_k, m_per_split, (unsigned)yb, (unsigned)xb, dbias, nullptr);
    return check_launch("gemm_tn(8w)");
  }
#endif
  if (conv)
    hipLaunchKernelGGL((gemm_tn_kernel<T, true>), dim3(tiles, splits), dim3(NT_THREADS), lds, s, (const T*)dY,
                       (const T*)X, dW, M, N, K, ldy, ldx, ldw, g, tiles_k, m_per_split, (unsigned)yb, (unsigned)xb, dbias, slabs);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<T, false>), dim3(tiles, splits), dim3(NT_THREADS), lds, s, (const T*)dY,
                       (const T*)X, dW, M, N, K, ldy, ldx, ldw, g, tiles_k, m_per_split, (unsigned)yb, (unsigned)xb, dbias, slabs);
  int rc = check_launch("gemm_tn");
  if (rc || !slabs) return rc;
  const long nk4 = (long)N * K / 4;
  long gr = (nk4 + 255) / 256;
  if (gr > 2048) gr = 2048;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((int)gr), dim3(256), 0, s, slabs, dW, splits, nk4, K, ldw);
  return check_launch("gemm_tn(reduce)");
}

}  // namespace spg

using namespace spg;

extern "C" int spg_gemm_nt(int dtype, const void* X, const void* W, void* C, void* C2, const float* bias,
                           const void* residual, const void* gelu_h, int M, int N, int K, int ldx, int ldc, int act,
                           int conv3x3, int B, int H, int Wd, int Ci, int cu_budget, spg_stream_t stream) {
  // the warm-up hint belongs to THIS call (whichever kernel it becomes: the families without the warm-up ignore it) and to no later one --
  // taken before any argument check, so a call that fails cannot leave a pointer behind for a launch made after its buffer is gone
  const void* const hint_ptr = t_hint_ptr;
  const unsigned hint_bytes = t_hint_bytes;
  t_hint_ptr = nullptr; t_hint_bytes = 0;
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(dtype == SPG_F32 || dtype == SPG_BF16, "gemm_nt: bad dtype %d", dtype);
  SPG_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  SPG_REQUIRE(K % vec == 0, "gemm_nt: K=%d must be a multiple of %d", K, vec);
  SPG_REQUIRE(ldc % 4 == 0 && ldc >= N, "gemm_nt: ldc=%d must be >=N and a multiple of 4", ldc);
  ConvGeom g{B, H, Wd, Ci};
  if (conv3x3) {
    SPG_REQUIRE((long)B * H * Wd == M && K == 9 * Ci && Ci % vec == 0, "gemm_nt: conv geometry mismatch (M=%d K=%d B=%d H=%d W=%d Ci=%d)", M, K, B, H, Wd, Ci);
  } else {
    SPG_REQUIRE(ldx % vec == 0 && ldx >= K, "gemm_nt: ldx=%d must be >=K and a multiple of %d", ldx, vec);
  }
  NtEpi epi{bias, residual, gelu_h, C2, act};
  hipStream_t s = (hipStream_t)stream;
  t_cur_pf = hint_ptr; t_cur_pf_bytes = hint_bytes;
  const int rc = dtype == SPG_BF16 ? launch_nt<bf16_t>(X, W, C, epi, M, N, K, ldx, ldc, conv3x3, g, s, cu_budget)
                                   : launch_nt<float>(X, W, C, epi, M, N, K, ldx, ldc, conv3x3, g, s, cu_budget);
  t_cur_pf = nullptr; t_cur_pf_bytes = 0;
  return rc;
}

extern "C" int spg_prefetch_hint(const void* next, long bytes) {
  SPG_REQUIRE(bytes >= 0 && (next != nullptr || bytes == 0), "prefetch_hint: bad range");
  SPG_REQUIRE(((uintptr_t)next & 3) == 0, "prefetch_hint: the range must start on a 4-byte boundary");
  t_hint_ptr = bytes > 0 ? next : nullptr;
  t_hint_bytes = bytes > 0x3FFFFFFFL ? 0x3FFFFFFFu : (unsigned)bytes;      // (a hint: anything beyond 1 GiB is simply not warmed)
  return SPG_OK;
}

#ifdef SPG_DEV_KERNELS
#include "dev/nt_chain_host.inc"
#endif
#ifdef SPG_DEV_KERNELS
extern "C" long spg_gemm_tn_group_workspace_bytes(void) {   // covers every kernel of the family
  const long a_ = (long)num_cus() * 2 * TNW_SLOT_FLOATS * (long)sizeof(float), b_ = tn_block_workspace_bytes(num_cus());
  return a_ > b_ ? a_ : b_;
}
#else
extern "C" long spg_gemm_tn_group_workspace_bytes(void) { return (long)num_cus() * 2 * TN_SLOT_FLOATS * (long)sizeof(float); }
#endif

extern "C" int spg_gemm_tn_group(int dtype, int njobs, const void* const* dY, const void* const* X, float* const* dW,
                                 float* const* dbias, int M, const int* N, const int* K, const int* ldy, const int* ldx,
                                 const int* ldw, void* workspace, long workspace_bytes, void* reduce_desc_out, int cu_budget,
                                 spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_BF16, "gemm_tn_group: bf16 only (dtype %d)", dtype);
  SPG_REQUIRE(njobs >= 1 && njobs <= TN_GROUP_MAX, "gemm_tn_group: 1..%d problems, got %d", TN_GROUP_MAX, njobs);
  SPG_REQUIRE(M > 0, "gemm_tn_group: empty M");
#ifdef SPG_DEV_KERNELS
  // opt-in (SPG_TN_BLOCK=1): one 256 x 192 block of dW per workgroup (tn_block.hip) for problem sets whose every N and K is a multiple of 192.
  // MEASURED: its main kernel takes 52.4 us on a stage-3 block against 65 for the tile kernel, but its partial blocks (49 MB of slabs)
  // need their own reduce launch per trunk block (14.5 us) where the tile kernel's boundary reduces are deferred and batched: the
  // train step is 23.40 ms with it and 23.39 ms without (same box).  Kept out of the product library.
  if (dev_env("SPG_TN_BLOCK", 0) != 0 && dev_env("SPG_TN_GROUP_WIDE", 0) == 0) {
    const int rcb = launch_tn_block_group(njobs, dY, X, dW, dbias, M, N, K, ldy, ldx, ldw, workspace, workspace_bytes, num_cus(cu_budget),
                                          (hipStream_t)stream);
    if (rcb != 1) {
      if (reduce_desc_out) { TnReduceDesc d; memset(&d, 0, sizeof(d)); memcpy(reduce_desc_out, &d, sizeof(d)); }   // reduced right here
      return rcb;
    }
  }
#endif
#ifdef SPG_DEV_KERNELS
#include "dev/tn_group_wide_dispatch.inc"
#endif  // SPG_DEV_KERNELS
  TnGroup g;
  long tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    SPG_REQUIRE(N[i] > 0 && K[i] > 0, "gemm_tn_group: empty problem %d", i);
    SPG_REQUIRE(N[i] % 8 == 0 && K[i] % 8 == 0 && ldy[i] % 8 == 0 && ldx[i] % 8 == 0 && ldw[i] % 4 == 0 && ldy[i] >= N[i] && ldx[i] >= K[i] &&
                    ldw[i] >= K[i],
                "gemm_tn_group: problem %d: N, K, ldy, ldx must be multiples of 8 (ldw of 4) and leading dimensions >= extents", i);
    SPG_REQUIRE((long)M * ldy[i] * 2 < 0xFFFFFFF0L && (long)M * ldx[i] * 2 < 0xFFFFFFF0L && (long)N[i] * ldw[i] * 4 < 0xFFFFFFF0L,
                "gemm_tn_group: problem %d: operand larger than 4 GiB", i);
    TnJob& jb = g.job[i];
    jb.dY = dY[i]; jb.X = X[i]; jb.dW = dW[i]; jb.dbias = dbias ? dbias[i] : nullptr;
    jb.N = N[i]; jb.K = K[i]; jb.ldy = ldy[i]; jb.ldx = ldx[i]; jb.ldw = ldw[i];
    jb.tiles_k = cdiv(K[i], 128); jb.tiles = cdiv(N[i], 128) * jb.tiles_k;
    jb.tile0 = (int)tiles;
    tiles += jb.tiles;
  }
  for (int i = njobs; i < TN_GROUP_MAX; ++i) g.job[i] = g.job[njobs - 1];
  const int S = cdiv(M, 64);
  SPG_REQUIRE(tiles * S < 0x7FFFFFFFL, "gemm_tn_group: too many steps");
  g.njobs = njobs; g.M = M; g.S = S; g.T = (int)tiles;
  const long total_steps = tiles * S;
  int G = total_steps < num_cus(cu_budget) ? (int)total_steps : num_cus(cu_budget);
#ifndef SPG_TN_ALIGN   // (A/B builds: -DSPG_TN_ALIGN=0 keeps one workgroup per CU whatever the tile count)
#define SPG_TN_ALIGN 1
#endif
  // Fewer tiles than CUs: every tile is cut into `parts` runs of steps.  With G = CUs the cuts fall at multiples of T S / G steps (stage 2:
  // 84 tiles x 288 steps / 256 = 94.5), so no two workgroups ever walk the same rows of M at the same time and every operand panel comes
  // from HBM once per TILE (measured: 583 MB fetched per stage-2 launch for 170 MB of operands, 98 us = 5.9 TB/s: HBM-bound).  With
  // G = T x parts (252) a run is exactly S / parts steps: the workgroups that hold the same part of neighbouring tiles start at the same
  // row, walk in lockstep and share the dY / X panels through their XCD's L2.  parts = 1 (243 tiles): whole tiles, no slabs at all.
  // Measured (tools/tn_group_bench.py, same box): a stage-2 trunk block 103.3 -> 70.3 us, the train step 22.9 -> 22.7 ms.
  g.parts = 0;
  if (SPG_TN_ALIGN && tiles <= G && total_steps >= G) {
    for (int parts = G / (int)tiles; parts >= 1; --parts) {
      if (S % parts != 0) continue;
      if ((long)tiles * parts * 100 >= (long)G * 85) { G = (int)tiles * parts; g.parts = SPG_TN_ALIGN == 2 ? 0 : parts; }   // (at least 85 % of the CUs keep a workgroup)
      break;
    }
  }
  g.W = (int)(tiles / G);
  g.RS = (int)((tiles - (long)g.W * G) * S);
  const long need = (long)G * 2 * TN_SLOT_FLOATS * (long)sizeof(float);
  SPG_REQUIRE(workspace && workspace_bytes >= need, "gemm_tn_group: workspace of %ld bytes needed (got %ld)", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDSG = 4 * 32768;
#ifdef SPG_DEV_KERNELS
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    attr = true;
  }
  const int dbgg = tn_group_v4() ? 0 : dev_env("SPG_TN_GROUP_DEBUG", 0);   // 8-wave kernel's ablations (wrong results by construction): 2 no MFMAs, 3 no fragment reads, 4 no fill, 5 stamps, 6 late DMA
  if (dbgg == 2) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t, 2>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else if (dbgg == 3) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t, 3>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else if (dbgg == 5) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t, 5>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else if (dbgg == 6) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t, 6>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else if (dbgg == 4) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group_kernel<bf16_t, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t, 4>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else
#endif
#ifdef SPG_DEV_KERNELS
  if (tn_group_v4() && dev_env("SPG_TN_GROUP_DEBUG", 0) == 5) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group4_kernel<bf16_t, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group4_kernel<bf16_t, 5>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else if (tn_group_v4() && dev_env("SPG_TN_GROUP_DEBUG", 0) == 2) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group4_kernel<bf16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group4_kernel<bf16_t, 2>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else
#endif
#ifdef SPG_DEV_KERNELS
  if (tn_group_v4() && dev_env("SPG_TN_GROUP_CW", 4) == 8) {   // A/B: eight multiplying waves (two per SIMD) + four loaders
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group4_kernel<bf16_t, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    hipLaunchKernelGGL((gemm_tn_group4_kernel<bf16_t, 0, 8>), dim3(G), dim3(768), LDSG, s, g, (float*)workspace, (unsigned)need);
  } else
#endif
  if (tn_group_v4()) {
    static bool attr4 = false;
    if (!attr4) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_group4_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
      attr4 = true;
    }
    hipLaunchKernelGGL((gemm_tn_group4_kernel<bf16_t>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
  }
#ifdef SPG_DEV_KERNELS
  else hipLaunchKernelGGL((gemm_tn_group_kernel<bf16_t>), dim3(G), dim3(512), LDSG, s, g, (float*)workspace, (unsigned)need);
#endif
  int rc = check_launch("gemm_tn_group");
  if (reduce_desc_out) {     // deferred: the caller collects descriptors and folds the slabs later (spg_gemm_tn_group_reduce_batch)
    TnReduceDesc d;
    d.g = g; d.G = (G < 2 || g.RS == 0) ? 0 : G; d.pad = 0;
    memcpy(reduce_desc_out, &d, sizeof(d));
    return rc;
  }
  if (rc || G < 2 || g.RS == 0) return rc;
  hipLaunchKernelGGL(tn_group_reduce_kernel, dim3(G - 1, 8), dim3(256), 0, s, g, (const float*)workspace, G);
  return check_launch("gemm_tn_group(reduce)");
}

extern "C" long spg_gemm_tn_blocks_count(int njobs, int M, const int* N, const int* K) { return tn_blocks_count(njobs, M, N, K); }
extern "C" int spg_num_cus(int cu_budget) { return num_cus(cu_budget); }

extern "C" int spg_gemm_tn_blocks(int dtype, int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias,
                                  int M, const int* N, const int* K, const int* ldy, const int* ldx, const int* ldw, int overwrite,
                                  float* sq_part, int cu_budget, spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_BF16, "gemm_tn_blocks: bf16 only (dtype %d)", dtype);
  SPG_REQUIRE(M > 0 && njobs >= 1, "gemm_tn_blocks: empty problem set");
  const int rc = launch_tn_blocks_direct(njobs, dY, X, dW, dbias, M, N, K, ldy, ldx, ldw, num_cus(cu_budget), (hipStream_t)stream, overwrite != 0,
                                         sq_part);
  SPG_REQUIRE(rc != 1, "gemm_tn_blocks: outside the kernel's domain (1..16 problems, M >= 256, every N and K a multiple of 192, leading "
                       "dimensions multiples of 8 (ldw: 4): ask spg_gemm_tn_blocks_count first)");
  return rc;
}

#ifdef SPG_DEV_KERNELS
extern "C" int spg_dev_tn_stamps(unsigned long long* out) {   // 256 workgroups x 8 waves x 4 segment sums (SPG_TN_GROUP_DEBUG=5)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::tn_stamp_sums), sizeof(unsigned long long) * 256 * 8 * 4) == hipSuccess ? 0 : -1;
}
#endif
extern "C" long spg_gemm_tn_group_desc_bytes(void) { return (long)sizeof(TnReduceDesc); }

extern "C" int spg_gemm_tn_group_reduce_batch(int n, const void* const* descs, const void* const* workspaces, spg_stream_t stream) {
  SPG_REQUIRE(n >= 1 && n <= TN_REDUCE_BATCH, "gemm_tn_group_reduce_batch: 1..%d groups, got %d", TN_REDUCE_BATCH, n);
  TnReduceBatch b;
  int maxG = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    TnReduceDesc d;
    memcpy(&d, descs[i], sizeof(d));
    if (d.G < 2) continue;                    // nothing was cut in that launch
    b.g[m] = d.g; b.slabs[m] = (const float*)workspaces[i]; b.G[m] = d.G;
    if (d.G > maxG) maxG = d.G;
    ++m;
  }
  if (m == 0) return SPG_OK;
  for (int i = m; i < TN_REDUCE_BATCH; ++i) { b.g[i] = b.g[0]; b.slabs[i] = b.slabs[0]; b.G[i] = 0; }
  b.n = m;
  #ifndef SPG_TN_REDUCE_YPARTS
#define SPG_TN_REDUCE_YPARTS 8
#endif
  hipLaunchKernelGGL(tn_group_reduce_batch_kernel, dim3(maxG - 1, SPG_TN_REDUCE_YPARTS, m), dim3(256), 0, (hipStream_t)stream, b);
  return check_launch("gemm_tn_group_reduce_batch");
}

extern "C" long spg_gemm_tn_workspace_bytes(int dtype, int M, int N, int K) {
  int splits, mps;
  if (dtype == SPG_BF16) tn_split_plan<bf16_t>(M, N, K, &splits, &mps);
  else tn_split_plan<float>(M, N, K, &splits, &mps);
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  int alt = num_cus() / tiles;                       // the persistent (dma) variant's plan
  if (alt > splits) splits = alt;
  if (splits < 1) splits = 1;   // the pipelined bf16 kernel always goes through a slab, also unsplit
  return (dtype == SPG_BF16 || splits > 1) ? (long)(splits + 1) * N * K * (long)sizeof(float) : 0;
}

extern "C" int spg_gemm_tn(int dtype, const void* dY, const void* X, float* dW, float* dbias, void* workspace, long workspace_bytes,
                           int M, int N, int K, int ldy, int ldx, int ldw, int conv3x3, int B, int H, int Wd, int Ci, int cu_budget,
                           spg_stream_t stream) {
  const int vec = dtype == SPG_BF16 ? 8 : 4;
  SPG_REQUIRE(dtype == SPG_F32 || dtype == SPG_BF16, "gemm_tn: bad dtype %d", dtype);
  SPG_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem");
  SPG_REQUIRE(N % vec == 0 && ldy % vec == 0 && K % vec == 0, "gemm_tn: N=%d ldy=%d K=%d must be multiples of %d", N, ldy, K, vec);
  ConvGeom g{B, H, Wd, Ci};
  if (conv3x3) {
    SPG_REQUIRE((long)B * H * Wd == M && K == 9 * Ci && Ci % vec == 0, "gemm_tn: conv geometry mismatch");
  } else {
    SPG_REQUIRE(ldx % vec == 0 && ldx >= K, "gemm_tn: bad ldx=%d", ldx);
  }
  hipStream_t s = (hipStream_t)stream;
  return dtype == SPG_BF16 ? launch_tn<bf16_t>(dY, X, dW, M, N, K, ldy, ldx, ldw, conv3x3, g, s, dbias, (float*)workspace, (size_t)workspace_bytes, cu_budget)
                           : launch_tn<float>(dY, X, dW, M, N, K, ldy, ldx, ldw, conv3x3, g, s, dbias, (float*)workspace, (size_t)workspace_bytes, cu_budget);
}

extern "C" int spg_pack_matrix(int dtype, const float* src, void* dst, int R, int C, int transpose, spg_stream_t stream) {
  SPG_REQUIRE(R > 0 && C > 0, "pack_matrix: empty");
  dim3 grid(cdiv(C, 32), cdiv(R, 32));
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(pack_matrix_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, transpose);
  else
    hipLaunchKernelGGL(pack_matrix_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, C, transpose);
  return check_launch("pack_matrix");
}

extern "C" int spg_pack_batch(int dtype, const void* jobs, int njobs, int total_tiles, spg_stream_t stream) {
  SPG_REQUIRE(njobs > 0 && total_tiles > 0, "pack_batch: empty job table");
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs, njobs);
  else
    hipLaunchKernelGGL(pack_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs, njobs);
  return check_launch("pack_batch");
}

extern "C" int spg_pack_conv3x3(int dtype, const float* src, void* dst_fwd, void* dst_dgrad, int Co, int Ci,
                                spg_stream_t stream) {
  const long n = (long)Co * Ci * 9;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst_fwd, (bf16_t*)dst_dgrad, Co, Ci);
  else
    hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst_fwd, (float*)dst_dgrad, Co, Ci);
  return check_launch("pack_conv3x3");
}

extern "C" int spg_unpack_conv3x3_grad(const float* packed, float* dst, int Co, int Ci, spg_stream_t stream) {
  const long n = (long)Co * Ci * 9;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpack_conv3x3_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, packed, dst, Co, Ci);
  return check_launch("unpack_conv3x3_grad");
}
