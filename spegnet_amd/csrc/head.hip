// Fused CFI / EFE / PED element kernels (HBM-bound; NHWC, 16-byte channel chunks, f32 arithmetic):
//
//   bn_apply_head     : y = relu(x*scale+shift) (optional store) and the 1x1 head logit  pred[m] = w . y[m,:] + b  in one pass
//                       (object_detection.py:150-155 EFE, :232-236 + :339 PED stage end) -- the head no longer re-reads y, and when
//                       nothing else needs y (last PED stage in training) it is never written.
//   ped_gather        : the input of a DecoderBlock's first conv in one pass:  pc = cat[ up2(act(x)), up_s(edge_features) ]
//                       with act = BN-apply + ReLU of the previous stage's raw conv output folded in (object_detection.py:219-232:
//                       F.interpolate x2, F.interpolate(edge -> size), torch.cat).  A thread owns one SOURCE pixel's 16-byte chunk:
//                       it loads the 3x3 source neighbourhood once (9 loads, 9 BN+ReLU) and writes the s x s output pixels that
//                       interpolate inside it (4 loads and 4 BN per output in the output-major form).
//   ped_gather_bwd    : exact adjoint in gather form (no atomics): a source pixel sums the <= 2s x 2s output pixels whose taps hit it.
//   bn_bwd_head_reduce/apply : BatchNorm backward whose incoming gradient is  dy = d_next (optional) + dpred[m]*w_head[c]  formed on the
//                       fly (the head's dx is rank one: it is never materialised), plus the head's own parameter gradients
//                       (dw_head[c] = sum_m dpred[m]*relu(bn(x))[m,c], db_head = sum_m dpred[m]) from the same pass.
//                       Deterministic: per-workgroup partials, fixed-order finish by the last workgroup (common.h).
#include "common.h"

namespace spg {

static inline int head_grid(long n_items) {
  long g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ------------------------------------------------------------------------------------------------------------------------------
template <typename T, bool WRITE_Y>
__global__ __launch_bounds__(256) void bn_apply_head_kernel(const T* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ y, T* __restrict__ pred, long M,
                                                            int C, int relu) {
  constexpr int VEC = ST<T>::VEC;
  const int lpp = C / VEC;  // lanes per pixel (power of two <= 64)
  const int ppw = 64 / lpp;
  const int lane = threadIdx.x & 63;
  const int sub = lane % lpp, pin = lane / lpp;
  const long wave_global = (blockIdx.x * 256L + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;
  float wv[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { wv[e] = w[sub * VEC + e]; sc[e] = ss[sub * VEC + e]; sh[e] = ss[C + sub * VEC + e]; }
  const float b0 = bias[0];
  for (long m0 = wave_global * ppw; m0 < M; m0 += nwaves * ppw) {
    const long m = m0 + pin;
    float s = 0.f;
    if (m < M) {
      float v[VEC];
      unpack16<T>(ld16(x + m * C + sub * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float o = v[e] * sc[e] + sh[e];
        v[e] = relu ? fmaxf(o, 0.f) : o;
      }
      if constexpr (WRITE_Y) {
        const u32x4 pk = pack16<T>(v);
        st16(y + m * C + sub * VEC, pk);
        unpack16<T>(pk, v);            // the head sees the stored (rounded) activation, as a separate pass over y would
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += v[e] * wv[e];
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && m < M) ST<T>::st(pred + m, s + b0);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// PyTorch bilinear, align_corners=False: src = max(0,(dst+0.5)*in/out-0.5); i0=floor; i1=min(i0+1,in-1)
__device__ __forceinline__ void bil_src_h(int dst, int in, int out, int& i0, int& i1, float& lam) {
  float src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
  src = fmaxf(src, 0.f);
  i0 = min((int)src, in - 1);
  i1 = min(i0 + 1, in - 1);
  lam = src - (float)i0;
}

struct GatherSrc {
  const void* x;          // [B, h, w, C] source
  const float* ss;        // scale/shift [2C] applied with ReLU before interpolation, or null (source already activated)
  int h, w, C, c0, s;     // c0: first output channel; s = output size / source size (2 or 4; the kernel is instantiated per s)
};

// the S x S outputs of source pixel (i, j), channel chunk ch, from its activated 3 x 3 neighbourhood (shared by the two kernels below: the
// same expressions, so the same bits)
template <typename T, int S>
__device__ __forceinline__ void gather_emit(const GatherSrc& g, const float (&nb)[3][3][ST<T>::VEC], T* __restrict__ y, int b, int i, int j, int ch,
                                            int H, int W, int ldy) {
  constexpr int VEC = ST<T>::VEC;
  float ly[S], lx[S];
#pragma unroll
  for (int o = 0; o < S; ++o) {
    int t0, t1;
    bil_src_h(i * S + o, g.h, H, t0, t1, ly[o]);
    bil_src_h(j * S + o, g.w, W, t0, t1, lx[o]);
  }
#pragma unroll
  for (int oy = 0; oy < S; ++oy) {
    const int r0 = (2 * oy < S) ? 0 : 1;
#pragma unroll
    for (int ox = 0; ox < S; ++ox) {
      const int q0 = (2 * ox < S) ? 0 : 1;
      const float w00 = (1.f - ly[oy]) * (1.f - lx[ox]), w01 = (1.f - ly[oy]) * lx[ox], w10 = ly[oy] * (1.f - lx[ox]), w11 = ly[oy] * lx[ox];
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        o[e] = w00 * nb[r0][q0][e] + w01 * nb[r0][q0 + 1][e] + w10 * nb[r0 + 1][q0][e] + w11 * nb[r0 + 1][q0 + 1][e];
      st16(y + (((long)b * H + (i * S + oy)) * W + (j * S + ox)) * ldy + g.c0 + ch * VEC, pack16<T>(o));
    }
  }
}

// Output pixel s*i + oy interpolates rows (i-1, i) of the source when oy < s/2 and rows (i, i+1) otherwise; with the neighbourhood
// clamped at the borders this holds there too (the clamped row repeats the border pixel, and PyTorch's lambda -- 0 at the low border,
// src - i0 at the high one -- is applied to two equal values exactly as here).  So tap selection is compile-time; only lambda is computed.
template <typename T, int S>
__device__ __forceinline__ void gather_one(const GatherSrc& g, T* __restrict__ y, long item, int H, int W, int ldy) {
  constexpr int VEC = ST<T>::VEC;
  const T* x = reinterpret_cast<const T*>(g.x);
  const int nch = g.C / VEC;
  const int ch = (int)(item % nch);
  long p = item / nch;
  const int j = (int)(p % g.w); p /= g.w;
  const int i = (int)(p % g.h);
  const int b = (int)(p / g.h);
  float sc[VEC], sh[VEC];
  const bool bn = g.ss != nullptr;
  if (bn) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = g.ss[ch * VEC + e]; sh[e] = g.ss[g.C + ch * VEC + e]; }
  }
  float nb[3][3][VEC];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = min(max(i + dy - 1, 0), g.h - 1);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = min(max(j + dx - 1, 0), g.w - 1);
      unpack16<T>(ld16(x + (((long)b * g.h + yy) * g.w + xx) * g.C + ch * VEC), nb[dy][dx]);
      if (bn) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) nb[dy][dx][e] = fmaxf(nb[dy][dx][e] * sc[e] + sh[e], 0.f);
      }
    }
  }
  gather_emit<T, S>(g, nb, y, b, i, j, ch, H, W, ldy);
}

template <typename T, int SE>
__global__ __launch_bounds__(256) void ped_gather_kernel(GatherSrc a, GatherSrc e, long items_a, long items_e, T* __restrict__ y, int H, int W,
                                                         int ldy) {
  const long total = items_a + items_e;
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    if (it < items_a) gather_one<T, 2>(a, y, it, H, W, ldy);
    else gather_one<T, SE>(e, y, it - items_a, H, W, ldy);
  }
}

// ---- LDS-tiled form.  The per-item kernel above loads a source pixel's 16-byte chunk NINE times (once per neighbour that interpolates
// from it) and applies the BatchNorm + ReLU nine times: its load + arithmetic half alone took 56 of 74 us on the 96 -> 192 stage with the
// stores compiled out, against 47 us for the stores with one load per item (tools/ped_probe.py).  Here a workgroup owns a 4 x 16-pixel
// source tile of 64 channels: the 6 x 18 halo tile is loaded once (16-byte loads, 128 contiguous bytes per pixel), activated once and
// kept in LDS as f32 (27 KiB); every thread then reads its 3 x 3 neighbourhood from LDS (conflict-free 16-byte reads) and emits its S x S
// outputs with the expressions of gather_one -- the same bits.
constexpr int PG_TH = 4, PG_TW = 16, PG_CCH = 8;
constexpr int PG_HALO = (PG_TH + 2) * (PG_TW + 2);
template <typename T, int S>
__device__ __forceinline__ void gather_tile(const GatherSrc& g, T* __restrict__ y, long tile_id, int H, int W, int ldy, float* __restrict__ tile) {
  constexpr int VEC = ST<T>::VEC;
  static_assert(VEC == 8, "the tile layout is written for 8-element chunks (bf16)");
  const T* x = reinterpret_cast<const T*>(g.x);
  const int nch = g.C / VEC, ncg = (nch + PG_CCH - 1) / PG_CCH;
  const int tw = (g.w + PG_TW - 1) / PG_TW, th = (g.h + PG_TH - 1) / PG_TH;
  const int cg = (int)(tile_id % ncg); long t = tile_id / ncg;
  const int tj = (int)(t % tw); t /= tw;
  const int ti = (int)(t % th);
  const int b = (int)(t / th);
  const int c = threadIdx.x & (PG_CCH - 1);                 // this thread's chunk inside the group, in both phases (256 % 8 == 0)
  const int ch = cg * PG_CCH + c;
  const bool ch_in = ch < nch;
  const bool bn = g.ss != nullptr;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = (bn && ch_in) ? g.ss[ch * VEC + e] : 1.f; sh[e] = (bn && ch_in) ? g.ss[g.C + ch * VEC + e] : 0.f; }
  // stage the halo tile: all of a thread's loads first, then activation + LDS writes
  constexpr int NIT = (PG_HALO * PG_CCH + 255) / 256;       // 4
  u32x4 raw[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int pix = (threadIdx.x >> 3) + k * 32;
    raw[k] = u32x4{0u, 0u, 0u, 0u};
    if (pix < PG_HALO && ch_in) {
      const int py = pix / (PG_TW + 2), px = pix - py * (PG_TW + 2);
      const int yy = min(max(ti * PG_TH + py - 1, 0), g.h - 1), xx = min(max(tj * PG_TW + px - 1, 0), g.w - 1);
      raw[k] = ld16(x + (((long)b * g.h + yy) * g.w + xx) * g.C + ch * VEC);
    }
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int pix = (threadIdx.x >> 3) + k * 32;
    if (pix < PG_HALO) {
      float v[VEC];
      unpack16<T>(raw[k], v);
      if (bn) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
      }
      float* d = tile + (pix * PG_CCH + c) * VEC;
      *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < (PG_TH * PG_TW * PG_CCH) / 256; ++k) {          // 2 source pixels per thread
    const int pix = (threadIdx.x >> 3) + k * 32;
    const int pi = pix / PG_TW, pj = pix - pi * PG_TW;
    const int i = ti * PG_TH + pi, j = tj * PG_TW + pj;
    if (i < g.h && j < g.w && ch_in) {
      float nb[3][3][VEC];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* sp = tile + (((pi + dy) * (PG_TW + 2) + (pj + dx)) * PG_CCH + c) * VEC;
          const f32x4 a = *reinterpret_cast<const f32x4*>(sp), bb = *reinterpret_cast<const f32x4*>(sp + 4);
          nb[dy][dx][0] = a[0]; nb[dy][dx][1] = a[1]; nb[dy][dx][2] = a[2]; nb[dy][dx][3] = a[3];
          nb[dy][dx][4] = bb[0]; nb[dy][dx][5] = bb[1]; nb[dy][dx][6] = bb[2]; nb[dy][dx][7] = bb[3];
        }
      gather_emit<T, S>(g, nb, y, b, i, j, ch, H, W, ldy);
    }
  }
}
template <typename T, int SE>
__global__ __launch_bounds__(256) void ped_gather_tiled_kernel(GatherSrc a, GatherSrc e, long tiles_a, T* __restrict__ y, int H, int W, int ldy) {
  __shared__ __attribute__((aligned(16))) float tile[PG_HALO * PG_CCH * 8];
  const long t = blockIdx.x;
  if (t < tiles_a) gather_tile<T, 2>(a, y, t, H, W, ldy, tile);
  else gather_tile<T, SE>(e, y, t - tiles_a, H, W, ldy, tile);
}

// adjoint (gather form): dx[b,i,j,c] (+)= sum over the output pixels whose taps include (i,j): exactly the 2S outputs
// Y in [S*i - S/2, S*i + 3S/2) per axis (the low border pixel also collects the clamped outputs, which that range contains); weights
// from the same bil_src_h arithmetic as the forward, so this is its exact adjoint, borders included.
template <typename T, int S>
__global__ __launch_bounds__(256) void ped_gather_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int h, int w, int C, int H,
                                                             int W, int ldy, int c0, int accumulate) {
  constexpr int VEC = ST<T>::VEC;
  constexpr int N = 2 * S;
  const int nch = C / VEC;
  // (32-bit index arithmetic: the 64-bit % and / of a long item index cost more VALU than the gather itself; the host checks the range)
  const unsigned total = (unsigned)B * h * w * nch;
  for (unsigned it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
    const int ch = (int)(it % (unsigned)nch);
    unsigned p = it / (unsigned)nch;
    const int j = (int)(p % (unsigned)w); p /= (unsigned)w;
    const int i = (int)(p % (unsigned)h);
    const int b = (int)(p / (unsigned)h);
    const int Y0 = S * i - S / 2, X0 = S * j - S / 2;
    float wy[N], wx[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int Y = Y0 + k, X = X0 + k;
      wy[k] = 0.f; wx[k] = 0.f;
      if (Y >= 0 && Y < H) {
        int y0, y1; float ly;
        bil_src_h(Y, h, H, y0, y1, ly);
        wy[k] = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
      }
      if (X >= 0 && X < W) {
        int x0, x1; float lx;
        bil_src_h(X, w, W, x0, x1, lx);
        wx[k] = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
      }
    }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // the candidates of KYB rows are requested together (all 16 of the x2 window, a row of 8 of the x4 window's 64), then added in the
    // same order: load-convert-add per candidate was one round trip each in the ISA (35 load groups for the x4 instance)
    constexpr int KYB = S == 2 ? 4 : 1;
#pragma unroll
    for (int ky0 = 0; ky0 < N; ky0 += KYB) {
      u32x4 raw[KYB][N];
#pragma unroll
      for (int kk = 0; kk < KYB; ++kk) {
        const int Y = min(max(Y0 + ky0 + kk, 0), H - 1);  // out-of-range candidates carry weight 0: clamp the address, keep the load
        const T* row = dy + (((long)b * H + Y) * W) * ldy + c0 + ch * VEC;
#pragma unroll
        for (int kx = 0; kx < N; ++kx) raw[kk][kx] = ld16(row + (long)min(max(X0 + kx, 0), W - 1) * ldy);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KYB; ++kk)
#pragma unroll
        for (int kx = 0; kx < N; ++kx) {
          float v[VEC];
          unpack16<T>(raw[kk][kx], v);
          const float wgt = wy[ky0 + kk] * wx[kx];
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += wgt * v[e];
        }
    }
    T* dst = dx + it * VEC;
    if (accumulate) {
      float o[VEC];
      unpack16<T>(ld16(dst), o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += o[e];
    }
    st16(dst, pack16<T>(acc));
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// BatchNorm backward with the 1x1 head's gradient formed on the fly.  dy'[m,c] = mask(m,c) * (dnext[m,c] + dpred[m]*hw[c]),
// mask = relu passed (recomputed from x).  Reduce: s0 = sum dy', s1 = sum dy'*xhat, s2 = sum dpred[m]*relu(bn(x)), sb = sum dpred.
// Grid (row blocks, 1, channel slabs) as colreduce_kernel (norm.hip); partial layout [slab][row block][4][SW].
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int HB_SLAB_CHUNKS = 16;
#ifndef SPG_HB_MAX_GX
#define SPG_HB_MAX_GX 512
#endif
#ifndef SPG_HB_ROWS_PER_THREAD
#define SPG_HB_ROWS_PER_THREAD 16
#endif
constexpr int HB_MAX_GX = SPG_HB_MAX_GX;

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_head_reduce_kernel(const T* __restrict__ dnext, const T* __restrict__ x, const T* __restrict__ dpred,
                                                                 const float* __restrict__ hw, const float* __restrict__ ss,
                                                                 const float* __restrict__ mi, float* __restrict__ sums /*[3C+1]*/, long M,
                                                                 int C, long rows_per_block, int nchs, float* __restrict__ part,
                                                                 unsigned* __restrict__ counters) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int rpar = 256 / nchs;
  const int chl = threadIdx.x % nchs, rl = threadIdx.x / nchs;
  const int slab = blockIdx.z, nslabs = gridDim.z;
  const int ch = slab * nchs + chl;
  const bool active = ch < nch && rl < rpar;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s0[VEC], s1[VEC], s2[VEC], sb = 0.f;
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], wv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    s0[e] = s1[e] = s2[e] = 0.f;
    const int c = active ? ch * VEC + e : 0;
    mu[e] = mi[c]; is[e] = mi[C + c]; sc[e] = ss[c]; sh[e] = ss[C + c]; wv[e] = hw[c];
  }
  if (active) {
    // four rows per trip, their loads requested before the first is used (x, the incoming gradient and the head's scalar of each row: written
    // row by row the compiler waited for every row's three loads in turn -- seen in the ISA as load groups of 3, 3, 3, 3; deeper: 8 rows 95.9
    // -> 93.2 us on the 384 x 384 x 64 stage, 59.6 -> 68.0 on 96 x 96 x 256).  A row past the block's end re-reads the block's first row, weight 0.
    const T* __restrict__ dsrc = dnext ? dnext : x;
    for (long rb = r0 + rl; rb < r1; rb += 4 * rpar) {
      u32x4 rx[4], rd[4];
      float dpv[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + (long)u * rpar;
        in[u] = r < r1;
        const long rc = in[u] ? r : r0 + rl;
        rx[u] = ld16(x + rc * C + ch * VEC);
        rd[u] = ld16(dsrc + rc * C + ch * VEC);
        dpv[u] = ST<T>::ld(dpred + rc);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!in[u]) continue;
        float xv[VEC], dv[VEC];
        unpack16<T>(rx[u], xv);
        if (dnext) unpack16<T>(rd[u], dv);
        else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) dv[e] = 0.f;
        }
        const float dp = dpv[u];
        if (chl == 0 && slab == 0) sb += dp;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float yv = xv[e] * sc[e] + sh[e];
          const bool on = yv > 0.f;
          const float d = on ? dv[e] + dp * wv[e] : 0.f;
          s0[e] += d;
          s1[e] += d * (xv[e] - mu[e]) * is[e];
          s2[e] += on ? dp * yv : 0.f;
        }
      }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[3][256 * 8];
  __shared__ __attribute__((aligned(16))) float fscr[256 * 4];
  __shared__ float redb[256];
  __shared__ unsigned s_last;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x * VEC + e] = s0[e]; red[1][threadIdx.x * VEC + e] = s1[e]; red[2][threadIdx.x * VEC + e] = s2[e];
  }
  redb[threadIdx.x] = sb;
  __syncthreads();
  const int SW = nchs * VEC;
  const int ldp = 3 * SW + 4;                          // partial row: [3][SW] sums + the head-bias sum (+ pad to 16 bytes)
  const int gx = gridDim.x;
  float* mypart = part + (((long)slab * gx) + blockIdx.x) * ldp;
  for (int c = threadIdx.x; c < SW; c += 256) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rpar; ++r) { t0 += red[0][r * SW + c]; t1 += red[1][r * SW + c]; t2 += red[2][r * SW + c]; }
    st_part(mypart + c, t0); st_part(mypart + SW + c, t1); st_part(mypart + 2 * SW + c, t2);
  }
  if (threadIdx.x < 4) {
    float t = 0.f;
    if (threadIdx.x == 0) for (int r = 0; r < 256; ++r) t += redb[r];
    st_part(mypart + 3 * SW + threadIdx.x, t);
  }
  if (!arrive_last(counters + slab, (unsigned)gx, &s_last)) return;
  const float* pbase = part + ((long)slab * gx) * ldp;
  const int ncols = min(SW, C - slab * SW);
  float* res = &red[0][0];
  finish_rows<256>(pbase, gx, ldp, res, fscr);
  for (int cl = threadIdx.x; cl < ncols; cl += 256) {
    sums[slab * SW + cl] = res[cl];
    sums[C + slab * SW + cl] = res[SW + cl];
    sums[2 * C + slab * SW + cl] = res[2 * SW + cl];
  }
  if (slab == 0 && threadIdx.x == 0) sums[3 * C] = res[3 * SW];
}

// dx = gamma*invstd*(dy' - s0/M - xhat*s1/M);  block 0 also: dgamma += s1, dbeta += s0, dw_head += s2, db_head += sb
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_head_apply_kernel(const T* __restrict__ dnext, const T* __restrict__ x, const T* __restrict__ dpred,
                                                                const float* __restrict__ hw, const float* __restrict__ ss,
                                                                const float* __restrict__ mi, const float* __restrict__ gamma,
                                                                const float* __restrict__ sums, T* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ dhw, float* __restrict__ dhb,
                                                                long M, int C, long rows_per_block) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int rpar = 256 / nch;
  const int ch = threadIdx.x % nch, rl = threadIdx.x / nch;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      dbeta[c] += sums[c];
      dgamma[c] += sums[C + c];
      dhw[c] += sums[2 * C + c];
    }
    if (threadIdx.x == 0) dhb[0] += sums[3 * C];
  }
  if (rl >= rpar) return;
  const float invM = 1.f / (float)M;
  float sc[VEC], sh[VEC], A[VEC], Bc[VEC], D[VEC], wv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int c = ch * VEC + e;
    sc[e] = ss[c]; sh[e] = ss[C + c]; wv[e] = hw[c];
    const float mu = mi[c], is = mi[C + c], g = gamma[c];
    const float s1 = sums[c] * invM, s2 = sums[C + c] * invM;
    A[e] = g * is;
    Bc[e] = -g * is * is * s2;
    D[e] = -g * is * s1 + g * is * is * mu * s2;
  }
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (long r = r0 + rl; r < r1; r += rpar) {
    float dv[VEC], xv[VEC];
    unpack16<T>(ld16(x + r * C + ch * VEC), xv);
    if (dnext) unpack16<T>(ld16(dnext + r * C + ch * VEC), dv);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dv[e] = 0.f;
    }
    const float dp = ST<T>::ld(dpred + r);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = (xv[e] * sc[e] + sh[e] > 0.f) ? dv[e] + dp * wv[e] : 0.f;
      dv[e] = A[e] * d + Bc[e] * xv[e] + D[e];
    }
    st16(dx + r * C + ch * VEC, pack16<T>(dv));
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// CFI fusion without the 2016-channel concat (feature_integration.py:229-241).  A 1x1 convolution commutes with bilinear
// interpolation (both are linear; the interpolation acts per channel, the convolution per pixel), so
//     conv1x1(cat[s2, up(s3), up(s4)]) = s2.W2^T + up(s3.W3^T) + up(s4.W4^T),      W = [W2 | W3 | W4] split by input channel:
// three GEMMs at each map's OWN resolution (4x fewer FLOPs than the GEMM over the upsampled concat, and no 74 MB concat buffer
// written and re-read) plus this one pass that adds the two upsampled products onto the first.
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cfi_combine_kernel(const T* __restrict__ y2, const T* __restrict__ y3, const T* __restrict__ y4,
                                                          T* __restrict__ out, int B, int H, int W, int h3, int w3, int h4, int w4, int C) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = (long)B * H * W * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    float o[VEC];
    unpack16<T>(ld16(y2 + i * VEC), o);
#pragma unroll
    for (int src = 0; src < 2; ++src) {
      const T* ys = src == 0 ? y3 : y4;
      const int h = src == 0 ? h3 : h4, w = src == 0 ? w3 : w4;
      int y0, y1, x0, x1; float ly, lx;
      bil_src_h(Y, h, H, y0, y1, ly);
      bil_src_h(X, w, W, x0, x1, lx);
      const T* base = ys + (long)b * h * w * C + ch * VEC;
      float a[VEC], bb[VEC], c[VEC], d[VEC];
      unpack16<T>(ld16(base + ((long)y0 * w + x0) * C), a);
      unpack16<T>(ld16(base + ((long)y0 * w + x1) * C), bb);
      unpack16<T>(ld16(base + ((long)y1 * w + x0) * C), c);
      unpack16<T>(ld16(base + ((long)y1 * w + x1) * C), d);
      const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] += w00 * a[e] + w01 * bb[e] + w10 * c[e] + w11 * d[e];
    }
    st16(out + i * VEC, pack16<T>(o));
  }
}

struct HbPlan { int nchs, nslabs, gx; long rpb; };
static inline HbPlan hb_plan(long rows, int nch) {
  HbPlan p;
  p.nchs = nch < HB_SLAB_CHUNKS ? nch : HB_SLAB_CHUNKS;
  p.nslabs = cdiv(nch, p.nchs);
  const int rpar = 256 / p.nchs;
  long want = rows / ((long)rpar * SPG_HB_ROWS_PER_THREAD);
  long cap = 2048 / p.nslabs;
  if (cap > HB_MAX_GX) cap = HB_MAX_GX;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  p.rpb = (rows + want - 1) / want;
  if (p.rpb < rpar) p.rpb = rpar;
  p.gx = cdiv(rows, p.rpb);
  return p;
}
static inline long bn_rows_per_block_h(long M, int nch) {
  const long rpar = 256 / nch;
  long rpb = (M + 4095) / 4096;
  if (rpb < rpar * 4) rpb = rpar * 4;
  return (rpb + rpar - 1) / rpar * rpar;
}

}  // namespace spg

using namespace spg;

static inline int vec_of_h(int dtype) { return dtype == SPG_BF16 ? 8 : 4; }

extern "C" int spg_bn_apply_head(int dtype, const void* x, const float* scale_shift, const float* w, const float* b, void* y, void* pred,
                                 long M, int C, int relu, spg_stream_t stream) {
  const int v = vec_of_h(dtype);
  const int lpp = C / v;
  SPG_REQUIRE(C % v == 0 && lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0, "bn_apply_head: C=%d/%d must be a power of two <= 64", C, v);
  SPG_REQUIRE(x && scale_shift && w && b && pred, "bn_apply_head: null argument");
  const int ppw = 64 / lpp;
  const int grid = head_grid((M + ppw - 1) / ppw * 64);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPG_BF16) {
    if (y) hipLaunchKernelGGL((bn_apply_head_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, scale_shift, w, b, (bf16_t*)y, (bf16_t*)pred, M, C, relu);
    else hipLaunchKernelGGL((bn_apply_head_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, scale_shift, w, b, (bf16_t*)nullptr, (bf16_t*)pred, M, C, relu);
  } else {
    if (y) hipLaunchKernelGGL((bn_apply_head_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)x, scale_shift, w, b, (float*)y, (float*)pred, M, C, relu);
    else hipLaunchKernelGGL((bn_apply_head_kernel<float, false>), dim3(grid), dim3(256), 0, s, (const float*)x, scale_shift, w, b, (float*)nullptr, (float*)pred, M, C, relu);
  }
  return check_launch("bn_apply_head");
}

extern "C" int spg_ped_gather(int dtype, const void* x, const float* x_scale_shift, int hx, int wx, int Cx, const void* edge, int he, int we,
                              int Ce, void* y, int B, int H, int W, spg_stream_t stream) {
  const int v = vec_of_h(dtype);
  SPG_REQUIRE(x && y && Cx % v == 0 && (Ce == 0 || (edge && Ce % v == 0)), "ped_gather: channel counts must be multiples of %d", v);
  SPG_REQUIRE(H == 2 * hx && W == 2 * wx, "ped_gather: the main input is upsampled exactly 2x (%dx%d -> %dx%d)", hx, wx, H, W);
  int se = 0;
  if (Ce) {
    se = H / he;
    SPG_REQUIRE((se == 2 || se == 4) && he * se == H && we * se == W, "ped_gather: edge features must be 1/2 or 1/4 of the output (%dx%d -> %dx%d)", he, we, H, W);
  }
  GatherSrc a{x, x_scale_shift, hx, wx, Cx, 0, 2};
  GatherSrc e{edge, nullptr, he, we, Ce, Cx, se ? se : 2};
  const long ia = (long)B * hx * wx * (Cx / v), ie = Ce ? (long)B * he * we * (Ce / v) : 0;
  const int grid = head_grid(ia + ie);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPG_BF16) {
#ifndef SPG_PED_TILED      // (tools/ A/B builds: 0 = the per-item kernel)
#define SPG_PED_TILED 1
#endif
    auto tiles = [](int B_, int h, int w, int C) { return (long)B_ * cdiv(h, PG_TH) * cdiv(w, PG_TW) * cdiv(C / 8, PG_CCH); };
    const long ta = tiles(B, hx, wx, Cx), te = Ce ? tiles(B, he, we, Ce) : 0;
    if (SPG_PED_TILED && ta + te < 0x7fffffffL) {
      if (se == 4) hipLaunchKernelGGL((ped_gather_tiled_kernel<bf16_t, 4>), dim3((unsigned)(ta + te)), dim3(256), 0, s, a, e, ta, (bf16_t*)y, H, W, Cx + Ce);
      else hipLaunchKernelGGL((ped_gather_tiled_kernel<bf16_t, 2>), dim3((unsigned)(ta + te)), dim3(256), 0, s, a, e, ta, (bf16_t*)y, H, W, Cx + Ce);
      return check_launch("ped_gather(tiled)");
    }
    if (se == 4) hipLaunchKernelGGL((ped_gather_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, s, a, e, ia, ie, (bf16_t*)y, H, W, Cx + Ce);
    else hipLaunchKernelGGL((ped_gather_kernel<bf16_t, 2>), dim3(grid), dim3(256), 0, s, a, e, ia, ie, (bf16_t*)y, H, W, Cx + Ce);
  } else {
    if (se == 4) hipLaunchKernelGGL((ped_gather_kernel<float, 4>), dim3(grid), dim3(256), 0, s, a, e, ia, ie, (float*)y, H, W, Cx + Ce);
    else hipLaunchKernelGGL((ped_gather_kernel<float, 2>), dim3(grid), dim3(256), 0, s, a, e, ia, ie, (float*)y, H, W, Cx + Ce);
  }
  return check_launch("ped_gather");
}

extern "C" int spg_ped_gather_bwd(int dtype, const void* dy, void* dx, int B, int h, int w, int C, int H, int W, int ldy, int c0, int accumulate,
                                  spg_stream_t stream) {
  const int v = vec_of_h(dtype);
  SPG_REQUIRE(C % v == 0 && ldy % v == 0 && c0 % v == 0, "ped_gather_bwd: alignment");
  const int sc = H / h;
  SPG_REQUIRE((sc == 2 || sc == 4) && h * sc == H && w * sc == W, "ped_gather_bwd: scale must be exactly 2 or 4");
  SPG_REQUIRE((long)B * H * W * ldy < 0x7fffffffL, "ped_gather_bwd: tensor too large for 32-bit index arithmetic");
  const int grid = head_grid((long)B * h * w * (C / v));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPG_BF16) {
    if (sc == 4) hipLaunchKernelGGL((ped_gather_bwd_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
    else hipLaunchKernelGGL((ped_gather_bwd_kernel<bf16_t, 2>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
  } else {
    if (sc == 4) hipLaunchKernelGGL((ped_gather_bwd_kernel<float, 4>), dim3(grid), dim3(256), 0, s, (const float*)dy, (float*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
    else hipLaunchKernelGGL((ped_gather_bwd_kernel<float, 2>), dim3(grid), dim3(256), 0, s, (const float*)dy, (float*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
  }
  return check_launch("ped_gather_bwd");
}

/* scratch of spg_bn_bwd_head: partial floats (any row count) and zeroed counters */
extern "C" long spg_bn_bwd_head_workspace_floats(int dtype, int C) {
  const int v = vec_of_h(dtype), nch = C / v, nchs = nch < HB_SLAB_CHUNKS ? nch : HB_SLAB_CHUNKS, nslabs = cdiv(nch, nchs);
  long cap = 2048 / nslabs;
  if (cap > HB_MAX_GX) cap = HB_MAX_GX;
  return cap * nslabs * (3L * nchs * v + 4);
}
extern "C" int spg_bn_bwd_head_counters(int dtype, int C) {
  const int v = vec_of_h(dtype), nch = C / v, nchs = nch < HB_SLAB_CHUNKS ? nch : HB_SLAB_CHUNKS;
  return cdiv(nch, nchs);
}

extern "C" int spg_bn_bwd_head(int dtype, const void* dnext, const void* x, const void* dpred, const float* head_w, const float* scale_shift,
                               const float* mean_invstd, const float* gamma, float* sums, void* dx, float* dgamma, float* dbeta,
                               float* dhead_w, float* dhead_b, long M, int C, float* red_ws, long red_ws_floats, unsigned* red_counters_,
                               spg_stream_t stream) {
  const int v = vec_of_h(dtype);
  SPG_REQUIRE(C % v == 0 && C / v <= 256, "bn_bwd_head: C=%d must be a multiple of %d and <= %d", C, v, 256 * v);
  SPG_REQUIRE(x && dpred && head_w && scale_shift && mean_invstd && gamma && sums && dx && dgamma && dbeta && dhead_w && dhead_b, "bn_bwd_head: null argument");
  const HbPlan p = hb_plan(M, C / v);
  const long need = (long)p.gx * p.nslabs * (3 * p.nchs * v + 4);
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= need, "bn_bwd_head: reduction workspace of %ld floats required, got %ld", need, red_ws_floats);
  hipStream_t s = (hipStream_t)stream;
  const long rpb = bn_rows_per_block_h(M, C / v);
  const int grid = cdiv(M, rpb);
  if (dtype == SPG_BF16) {
    hipLaunchKernelGGL(bn_bwd_head_reduce_kernel<bf16_t>, dim3(p.gx, 1, p.nslabs), dim3(256), 0, s, (const bf16_t*)dnext, (const bf16_t*)x, (const bf16_t*)dpred,
                       head_w, scale_shift, mean_invstd, sums, M, C, p.rpb, p.nchs, red_ws, red_counters_);
    hipLaunchKernelGGL(bn_bwd_head_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dnext, (const bf16_t*)x, (const bf16_t*)dpred, head_w,
                       scale_shift, mean_invstd, gamma, sums, (bf16_t*)dx, dgamma, dbeta, dhead_w, dhead_b, M, C, rpb);
  } else {
    hipLaunchKernelGGL(bn_bwd_head_reduce_kernel<float>, dim3(p.gx, 1, p.nslabs), dim3(256), 0, s, (const float*)dnext, (const float*)x, (const float*)dpred,
                       head_w, scale_shift, mean_invstd, sums, M, C, p.rpb, p.nchs, red_ws, red_counters_);
    hipLaunchKernelGGL(bn_bwd_head_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dnext, (const float*)x, (const float*)dpred, head_w,
                       scale_shift, mean_invstd, gamma, sums, (float*)dx, dgamma, dbeta, dhead_w, dhead_b, M, C, rpb);
  }
  return check_launch("bn_bwd_head");
}

extern "C" int spg_cfi_combine(int dtype, const void* y2, const void* y3, const void* y4, void* out, int B, int H, int W, int h3, int w3,
                               int h4, int w4, int C, spg_stream_t stream) {
  const int v = vec_of_h(dtype);
  SPG_REQUIRE(y2 && y3 && y4 && out && C % v == 0, "cfi_combine: C=%d must be a multiple of %d", C, v);
  SPG_REQUIRE(h3 <= H && w3 <= W && h4 <= H && w4 <= W && h3 > 0 && h4 > 0, "cfi_combine: the coarser maps are upsampled to %dx%d", H, W);
  const int grid = head_grid((long)B * H * W * (C / v));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(cfi_combine_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y2, (const bf16_t*)y3, (const bf16_t*)y4, (bf16_t*)out, B, H, W, h3, w3, h4, w4, C);
  else hipLaunchKernelGGL(cfi_combine_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)y2, (const float*)y3, (const float*)y4, (float*)out, B, H, W, h3, w3, h4, w4, C);
  return check_launch("cfi_combine");
}
