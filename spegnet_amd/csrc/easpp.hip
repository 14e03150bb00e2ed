// e-ASPP middle (feature_integration.py:397-412) as branch-batched kernels.  The four dilated depth-wise branches live in ONE tensor
// dcat [M, 4C] whose channel order IS the reference's branch-major concat (torch.cat([b0,b1,b2,b3,global]) :411), so
//   * one launch computes all four dilated 3x3 convolutions (dwconv4),
//   * one deterministic reduction gives the batch statistics of all four BatchNorms (spg_bn_stats_finalize with 4 parameter groups),
//   * the grouped 1x1 fusion conv reads 5 CONSECUTIVE channels of dcat per group (the branch-major quirk) and applies the
//     branches' BN + ReLU on the fly (easpp_fuse_bn): the four activated branch tensors are never written,
//   * the global branch (GAP -> 1x1 -> BN over B values -> ReLU -> broadcast) is one single-workgroup kernel in each direction,
//   * backward mirrors it: the fusion conv's gradient w.r.t. dcat is formed on the fly inside the BatchNorm backward of the four
//     branch BNs (reduce + apply over [M, 4C]), then one launch each for the depth-wise weight gradients and the input gradient
//     (all four branches + the global-average-pool adjoint summed in registers: no accumulate passes).
// ~34 launches forward / ~45 backward become 8 / 10.  All reductions are deterministic (common.h).
#include "common.h"

namespace spg {

static inline int ea_grid(long n_items) {
  long g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

struct Dw4 { const float* w[4]; int dil[4]; };     // weights f32 [C][9] per branch

// ---- forward: y[p][br*C + c] = sum_t w_br[c][t] * x[p + t*dil_br][c];  blockIdx.y = branch, weights of the thread's chunk in registers
template <typename T>
__global__ __launch_bounds__(256) void dwconv4_kernel(const T* __restrict__ x, Dw4 d, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int VEC = ST<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [9][C] of this branch
  const int nch = C / VEC;
  const int br = blockIdx.y, dil = d.dil[br];
  for (int i = threadIdx.x; i < 9 * C; i += 256) { const int t = i / C, c = i - t * C; wl[i] = d.w[br][c * 9 + t]; }
  __syncthreads();
  const long total = (long)B * H * W * nch;
  const int ch = threadIdx.x % nch;                 // loop invariant: 256 and the grid stride are multiples of nch (host checks)
  float wr[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) wr[t][e] = wl[t * C + ch * VEC + e];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long p = i / nch;
    const long pix = p;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // The nine taps are loaded UNCONDITIONALLY (a tap outside the image reads the centre pixel and is weighted 0) and all before the
    // first use: a load inside `if (inside)` is waited for inside that branch, nine dependent memory round trips per pixel -- what these
    // kernels' time was (forward 31 us, weight gradient 61 us for 24 MB).
    u32x4 raw[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = Y + (t / 3 - 1) * dil, xx = X + (t % 3 - 1) * dil;
      ok[t] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const long q = ok[t] ? (((long)b * H + yy) * W + xx) : pix;
      raw[t] = ld16(x + q * C + ch * VEC);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v[VEC];
      unpack16<T>(raw[t], v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = ok[t] ? fmaf(wr[t][e], v[e], acc[e]) : acc[e];
    }
    st16(y + pix * 4 * C + br * C + ch * VEC, pack16<T>(acc));
  }
}

// ---- input gradient of all four branches + the GAP adjoint: dx[p][c] = gadd[b][c] + sum_br sum_t w_br[c][8-t] * dy[p + t*dil_br][br*C + c]
// weights staged once per workgroup in LDS as [br][t][C] (a thread reads its chunk's VEC weights per tap as 16-byte reads)
template <typename T>
__global__ __launch_bounds__(256) void dwconv4_dgrad_kernel(const T* __restrict__ dy, Dw4 d, const float* __restrict__ gadd, T* __restrict__ dx,
                                                            int B, int H, int W, int C) {
  constexpr int VEC = ST<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [4][9][C]
  for (int i = threadIdx.x; i < 4 * 9 * C; i += 256) {
    const int br = i / (9 * C), r = i - br * 9 * C, t = r / C, c = r - t * C;
    wl[i] = d.w[br][c * 9 + (8 - t)];                           // flipped: the adjoint of the correlation
  }
  __syncthreads();
  const int nch = C / VEC;
  const long total = (long)B * H * W * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const long pix = p;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = gadd ? gadd[(long)b * C + ch * VEC + e] : 0.f;
#pragma unroll
    for (int br = 0; br < 4; ++br) {
      const int dil = d.dil[br];
      u32x4 raw[9];     // (a branch's nine taps in flight together, see dwconv4_kernel)
      bool ok[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = Y + (t / 3 - 1) * dil, xx = X + (t % 3 - 1) * dil;
        ok[t] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const long q = ok[t] ? (((long)b * H + yy) * W + xx) : pix;
        raw[t] = ld16(dy + q * 4 * C + br * C + ch * VEC);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v[VEC];
        unpack16<T>(raw[t], v);
        const float* wp = wl + (br * 9 + t) * C + ch * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = ok[t] ? fmaf(wp[e], v[e], acc[e]) : acc[e];
      }
    }
    st16(dx + i * VEC, pack16<T>(acc));
  }
}

// ---- depth-wise weight gradients of all four branches: dw_br[c][t] += sum_p dy[p][br*C + c] * x[p + t*dil_br][c]; blockIdx.y = branch
template <typename T>
__global__ __launch_bounds__(256) void dwconv4_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, Dw4 d, float* const dw0,
                                                            float* const dw1, float* const dw2, float* const dw3, int B, int H, int W, int C,
                                                            long pix_per_block, float* __restrict__ part, unsigned* __restrict__ counters) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int ppar = 256 / nch;
  const int ch = threadIdx.x % nch, pl = threadIdx.x / nch;
  const int br = blockIdx.y, dil = d.dil[br];
  const long npix = (long)B * H * W;
  const long p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  float acc[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
  if (pl < ppar) {
    for (long p = p0 + pl; p < p1; p += ppar) {
      const int X = (int)(p % W);
      const int Y = (int)((p / W) % H);
      const u32x4 rdy = ld16(dy + p * 4 * C + br * C + ch * VEC);
      u32x4 raw[9];     // (the nine taps in flight together, see dwconv4_kernel)
      bool ok[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = Y + (t / 3 - 1) * dil, xx = X + (t % 3 - 1) * dil;
        ok[t] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const long q = ok[t] ? p + (long)(t / 3 - 1) * dil * W + (t % 3 - 1) * dil : p;
        raw[t] = ld16(x + q * C + ch * VEC);
      }
      __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise pairs each tap's conversion with its load: groups of two in the ISA)
      float dv[VEC];
      unpack16<T>(rdy, dv);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v[VEC];
        unpack16<T>(raw[t], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = ok[t] ? fmaf(dv[e], v[e], acc[t][e]) : acc[t][e];
      }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[256 * 8];
  __shared__ __attribute__((aligned(16))) float fscr[256 * 4];
  __shared__ unsigned s_last;
  const int gx = gridDim.x;
  float* mypart = part + ((long)br * gx + blockIdx.x) * 9 * C;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = acc[t][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      for (int r = 0; r < ppar; ++r) s += red[r * nch * VEC + c];
      st_part(mypart + (long)c * 9 + t, s);
    }
  }
  if (!arrive_last(counters + br, (unsigned)gx, &s_last)) return;
  float* dw = br == 0 ? dw0 : (br == 1 ? dw1 : (br == 2 ? dw2 : dw3));
  finish_rows<256>(part + (long)br * gx * 9 * C, gx, 9 * C, red, fscr);      // 9C <= 2048 floats (host checks)
  for (int i = threadIdx.x; i < 9 * C; i += 256) dw[i] += red[i];
}

// ---- grouped 1x1 fusion over the branch-major concat with the branch BN + ReLU applied on the fly:
// y[p][g] = sum_j w[g][j] * cat[p][5g+j],  cat = [relu(dcat*scale+shift) (4C channels) | glob[b] (C channels, already activated)]
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_bn_kernel(const T* __restrict__ dcat, const float* __restrict__ ss, const float* __restrict__ glob,
                                                            const float* __restrict__ w, T* __restrict__ y, long HW, int C, long total) {
  const int C4 = 4 * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i % C);
    const long p = i / C;
    const long b = p / HW;
    // (the five taps' operands requested together -- both sources with clamped indices, the one that applies chosen afterwards: a load in each
    // arm of `if (cc < 4C)` was waited for inside its arm, five round trips per output)
    float xv[5], sc[5], sh[5], gv[5], wv[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int cc = 5 * g + j;
      const int cd = cc < C4 ? cc : 0, cg = cc < C4 ? 0 : cc - C4;
      xv[j] = ST<T>::ld(dcat + p * C4 + cd); sc[j] = ss[cd]; sh[j] = ss[C4 + cd];
      gv[j] = glob[b * C + cg]; wv[j] = w[g * 5 + j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int cc = 5 * g + j;
      const float v = cc < C4 ? fmaxf(xv[j] * sc[j] + sh[j], 0.f) : gv[j];
      s += wv[j] * v;
    }
    ST<T>::st(y + i, s);
  }
}

// ---- backward of the fusion conv + the four branch BatchNorms in one reduce + apply pair over [M, 4C].
// incoming gradient of concat channel cc (< 4C):  dy'[p][cc] = mask * w[cc/5][cc%5] * dfu[p][cc/5],  mask = relu passed.
// reduce: s0 = sum dy', s1 = sum dy'*xhat (BN), s2 = sum dfu[p][g] * relu(bn(x)) (the fusion conv's weight gradient)
constexpr int EF_SLAB_CHUNKS = 16;
constexpr int EF_MAX_GX = 512;
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_bn_bwd_reduce_kernel(const T* __restrict__ dfu, const T* __restrict__ dcat,
                                                                       const float* __restrict__ w, const float* __restrict__ ss,
                                                                       const float* __restrict__ mi, float* __restrict__ sums /*[3][4C]*/, long M,
                                                                       int C, long rows_per_block, int nchs, float* __restrict__ part,
                                                                       unsigned* __restrict__ counters) {
  constexpr int VEC = ST<T>::VEC;
  const int C4 = 4 * C;
  const int nch = C4 / VEC;
  const int rpar = 256 / nchs;
  const int chl = threadIdx.x % nchs, rl = threadIdx.x / nchs;
  const int slab = blockIdx.z, nslabs = gridDim.z;
  const int ch = slab * nchs + chl;
  const bool active = ch < nch && rl < rpar;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s0[VEC], s1[VEC], s2[VEC], mu[VEC], is[VEC], sc[VEC], sh[VEC], wv[VEC];
  int gi[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    s0[e] = s1[e] = s2[e] = 0.f;
    const int cc = active ? ch * VEC + e : 0;
    mu[e] = mi[cc]; is[e] = mi[C4 + cc]; sc[e] = ss[cc]; sh[e] = ss[C4 + cc]; wv[e] = w[cc]; gi[e] = cc / 5;   // w[g][j] == w[cc] (cc = 5g+j)
  }
  if (active) {
    for (long r = r0 + rl; r < r1; r += rpar) {
      float xv[VEC];
      unpack16<T>(ld16(dcat + r * C4 + ch * VEC), xv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float df = ST<T>::ld(dfu + r * C + gi[e]);
        const float yv = xv[e] * sc[e] + sh[e];
        const bool on = yv > 0.f;
        const float dd = on ? df * wv[e] : 0.f;
        s0[e] += dd;
        s1[e] += dd * (xv[e] - mu[e]) * is[e];
        s2[e] += on ? df * yv : 0.f;
      }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[3][256 * 8];
  __shared__ __attribute__((aligned(16))) float fscr[256 * 4];
  __shared__ unsigned s_last;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][threadIdx.x * VEC + e] = s0[e]; red[1][threadIdx.x * VEC + e] = s1[e]; red[2][threadIdx.x * VEC + e] = s2[e];
  }
  __syncthreads();
  const int SW = nchs * VEC;
  const int gx = gridDim.x;
  float* mypart = part + (((long)slab * gx) + blockIdx.x) * (3 * SW);
  for (int c = threadIdx.x; c < SW; c += 256) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rpar; ++r) { t0 += red[0][r * SW + c]; t1 += red[1][r * SW + c]; t2 += red[2][r * SW + c]; }
    st_part(mypart + c, t0); st_part(mypart + SW + c, t1); st_part(mypart + 2 * SW + c, t2);
  }
  if (!arrive_last(counters + slab, (unsigned)gx, &s_last)) return;
  const float* pbase = part + ((long)slab * gx) * (3 * SW);
  const int ncols = min(SW, C4 - slab * SW);
  float* res = &red[0][0];
  finish_rows<256>(pbase, gx, 3 * SW, res, fscr);
  for (int cl = threadIdx.x; cl < ncols; cl += 256) {
    sums[slab * SW + cl] = res[cl];
    sums[C4 + slab * SW + cl] = res[SW + cl];
    sums[2 * C4 + slab * SW + cl] = res[2 * SW + cl];
  }
}

struct Bn4 { const float* gamma[4]; float* dgamma[4]; float* dbeta[4]; };
// apply: d_dcat = gamma*invstd*(dy' - s0/M - xhat*s1/M); block 0 also: dgamma_br += s1, dbeta_br += s0, dw_fuse[cc] += s2
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_bn_bwd_apply_kernel(const T* __restrict__ dfu, const T* __restrict__ dcat, const float* __restrict__ w,
                                                                      const float* __restrict__ ss, const float* __restrict__ mi, Bn4 bn,
                                                                      const float* __restrict__ sums, T* __restrict__ ddcat, float* __restrict__ dwf,
                                                                      long M, int C, long total) {
  constexpr int VEC = ST<T>::VEC;
  const int C4 = 4 * C;
  const int nch = C4 / VEC;
  if (blockIdx.x == 0) {
    for (int cc = threadIdx.x; cc < C4; cc += 256) {
      const int br = cc / C, c = cc - br * C;
      bn.dbeta[br][c] += sums[cc];
      bn.dgamma[br][c] += sums[C4 + cc];
      dwf[cc] += sums[2 * C4 + cc];
    }
  }
  const float invM = 1.f / (float)M;
  const int ch = threadIdx.x % nch;                  // loop invariant (256 % nch == 0, host checks)
  float sc[VEC], sh[VEC], A[VEC], Bc[VEC], D[VEC], wv[VEC];
  int gi[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int cc = ch * VEC + e;
    const int br = cc / C;
    sc[e] = ss[cc]; sh[e] = ss[C4 + cc]; wv[e] = w[cc]; gi[e] = cc / 5;
    const float mu = mi[cc], is = mi[C4 + cc], g = bn.gamma[br][cc - br * C];
    const float s1 = sums[cc] * invM, s2 = sums[C4 + cc] * invM;
    A[e] = g * is;
    Bc[e] = -g * is * is * s2;
    D[e] = -g * is * s1 + g * is * is * mu * s2;
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    float xv[VEC], o[VEC];
    unpack16<T>(ld16(dcat + i * VEC), xv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float df = ST<T>::ld(dfu + r * C + gi[e]);
      const float dd = (xv[e] * sc[e] + sh[e] > 0.f) ? df * wv[e] : 0.f;
      o[e] = A[e] * dd + Bc[e] * xv[e] + D[e];
    }
    st16(ddcat + i * VEC, pack16<T>(o));
  }
}

// ---- global branch, forward (one workgroup): gm = gsum/HW; gl0 = gm . Wg^T; BN over the B values of each channel (train: batch
// statistics + running update; eval: running); glob = relu(.)
__global__ __launch_bounds__(1024) void easpp_global_fwd_kernel(const float* __restrict__ gsum, const float* __restrict__ Wg,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, long long* __restrict__ nbt, float* __restrict__ gm,
                                                                float* __restrict__ gl0, float* __restrict__ glob, float* __restrict__ ss,
                                                                float* __restrict__ mi, int B, int C, float inv_hw, float eps, float momentum,
                                                                int training, int stage_w) {
  extern __shared__ float sm[];   // gm [B][C], then gl0 [B][C], then (stage_w) Wg as [C][C + 4]
  float* sl = sm + B * C;
  float* sW = sl + B * C;
  const int nt = (int)blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
  if (stage_w) {
    // The cold weight matrix arrives with the pooled sums in ONE memory round trip (16-byte loads by every thread), rows padded by four
    // floats: a thread then owns one output (image, channel) and walks its weight row and the image's pooled vector with 16-byte LDS
    // reads (rows of consecutive channels start 4 banks apart: conflict-free; the pooled vector is a broadcast) -- no wave reductions.
    // The wave-per-output form below spent this single-workgroup kernel in them: 8 outputs x 8 images x 6 cross-lane steps per wave,
    // 16 waves on one LDS pipe, after one memory round trip per output (32 us; 30 with the weights staged but the reductions kept).
    const int CP = C + 4;
    for (int i = threadIdx.x * 4; i < C * C; i += nt * 4) {
      const int o = i / C, k = i - o * C;
      *reinterpret_cast<float4*>(sW + o * CP + k) = *reinterpret_cast<const float4*>(Wg + i);
    }
    for (int i = threadIdx.x; i < B * C; i += nt) { sm[i] = gsum[i] * inv_hw; gm[i] = sm[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < B * C; i += nt) {
      const int b = i / C, o = i - b * C;
      const float* wr = sW + o * CP;
      const float* gv = sm + b * C;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int k = 0; k < C; k += 4) {
        const float4 w = *reinterpret_cast<const float4*>(wr + k);
        const float4 g = *reinterpret_cast<const float4*>(gv + k);
        s0 += w.x * g.x; s1 += w.y * g.y; s2 += w.z * g.z; s3 += w.w * g.w;
      }
      const float t = (s0 + s1) + (s2 + s3);
      sl[i] = t; gl0[i] = t;
    }
  } else {
  for (int i = threadIdx.x; i < B * C; i += nt) { sm[i] = gsum[i] * inv_hw; gm[i] = sm[i]; }
  __syncthreads();
  // 1x1 conv: a wave per output channel, lanes stride the input channels (coalesced weight rows), 8 images per pass
  for (int o = wave; o < C; o += nw) {
    for (int b0 = 0; b0 < B; b0 += 8) {
      float s[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = 0.f;
      for (int k = lane; k < C; k += 64) {
        const float w = Wg[(long)o * C + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (b0 + j < B) s[j] += w * sm[(b0 + j) * C + k];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum(s[j]);
        if (lane == 0 && b0 + j < B) { sl[(b0 + j) * C + o] = t; gl0[(b0 + j) * C + o] = t; }
      }
    }
  }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += nt) {
    float mu = 0.f;
    for (int b = 0; b < B; ++b) mu += sl[b * C + o];
    float var;
    if (training) {
      mu /= (float)B;
      float q = 0.f;
      for (int b = 0; b < B; ++b) { const float dlt = sl[b * C + o] - mu; q += dlt * dlt; }
      var = q / (float)B;
      if (rmean) {
        rmean[o] = (1.f - momentum) * rmean[o] + momentum * mu;
        rvar[o] = (1.f - momentum) * rvar[o] + momentum * (B > 1 ? var * ((float)B / (float)(B - 1)) : var);
      }
    } else {
      mu = rmean[o]; var = rvar[o];
    }
    const float is = rsqrtf(var + eps), sc = gamma[o] * is, sh = beta[o] - mu * sc;
    ss[o] = sc; ss[C + o] = sh; mi[o] = mu; mi[C + o] = is;
    for (int b = 0; b < B; ++b) glob[b * C + o] = fmaxf(sl[b * C + o] * sc + sh, 0.f);
  }
  if (training && threadIdx.x == 0 && nbt) nbt[0] += 1;
}

// ---- global branch, backward (one workgroup).  S[b][g] = sum_p dfu[b,p,g] (per-image column sums of the fusion conv's output gradient).
// concat channel cc = 4C + ch (ch < C) belongs to group g = cc/5, tap j = cc%5:  dglob[b][ch] = w[cc]*S[b][g],  dw_fuse[cc] += sum_b glob[b][ch]*S[b][g];
// then BN (over B) + ReLU backward, the 1x1 conv's weight gradient and the GAP adjoint  gadd[b][k] = (sum_o d_gl0[b][o]*Wg[o][k]) / HW.
__global__ __launch_bounds__(1024) void easpp_global_bwd_kernel(const float* __restrict__ S, const float* __restrict__ glob, const float* __restrict__ gl0,
                                                               const float* __restrict__ gm, const float* __restrict__ wf, const float* __restrict__ Wg,
                                                               const float* __restrict__ gamma, const float* __restrict__ mi, float* __restrict__ dwf,
                                                               float* __restrict__ dWg, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ gadd, int B, int C, float inv_hw, int training) {
  extern __shared__ float sm[];   // dgl0 [B][C], then gm [B][C] (staged: the weight-gradient loop below read it from global memory, B loads per element)
  const int C4 = 4 * C, nt = (int)blockDim.x;
  float* gml = sm + B * C;
  for (int i = threadIdx.x; i < B * C; i += nt) gml[i] = gm[i];
  for (int ch = threadIdx.x; ch < C; ch += nt) {
    const int cc = C4 + ch, g = cc / 5;
    const float wv = wf[cc];
    float dwsum = 0.f, s0 = 0.f, s1 = 0.f;
    const float mu = mi[ch], is = mi[C + ch];
    for (int b0 = 0; b0 < B; b0 += 8) {        // eight images' operands requested together (they were one round trip per image), used in image order
      float sgv[8], glv[8], g0v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u < B ? b0 + u : b0;
        sgv[u] = S[b * C + g]; glv[u] = glob[b * C + ch]; g0v[u] = gl0[b * C + ch];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u;
        if (b < B) {
          const float sg = sgv[u];
          dwsum += glv[u] * sg;
          const float dy = glv[u] > 0.f ? wv * sg : 0.f;      // through the ReLU
          sm[b * C + ch] = dy;
          s0 += dy;
          s1 += dy * (g0v[u] - mu) * is;
        }
      }
    }
    dwf[cc] += dwsum;
    dbeta[ch] += s0;
    dgamma[ch] += s1;
    const float gmm = gamma[ch];
    for (int b0 = 0; b0 < B; b0 += 8) {
      float g0v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) g0v[u] = gl0[(b0 + u < B ? b0 + u : b0) * C + ch];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u;
        if (b < B) {
          const float dy = sm[b * C + ch];
          // training: full BN backward over the B samples; eval: the affine map only
          sm[b * C + ch] = training ? gmm * is * (dy - s0 / (float)B - (g0v[u] - mu) * is * s1 / (float)B) : gmm * is * dy;
        }
      }
    }
  }
  __syncthreads();
  // (read-modify-write of the weight gradient, eight elements per thread at a time: the old values are loaded TOGETHER -- written as
  // `dWg[i] += s` in a loop the sixteen updates of a thread were sixteen dependent memory round trips, half of this kernel's 34 us)
  for (int i0 = threadIdx.x; i0 < C * C; i0 += nt * 8) {
    float old[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = i0 + j * nt; old[j] = i < C * C ? dWg[i] : 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * nt;
      if (i < C * C) {
        const int o = i / C, k = i - o * C;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sm[b * C + o] * gml[b * C + k];
        dWg[i] = old[j] + s;
      }
    }
  }
  for (int i = threadIdx.x; i < B * C; i += nt) {
    const int b = i / C, k = i - b * C;
    float s = 0.f;
    for (int o = 0; o < C; ++o) s += sm[b * C + o] * Wg[(long)o * C + k];
    gadd[i] = s * inv_hw;
  }
}

struct EfPlan { int nchs, nslabs, gx; long rpb; };
static inline EfPlan ef_plan(long rows, int nch) {
  EfPlan p;
  p.nchs = nch < EF_SLAB_CHUNKS ? nch : EF_SLAB_CHUNKS;
  p.nslabs = cdiv(nch, p.nchs);
  const int rpar = 256 / p.nchs;
  long want = rows / ((long)rpar * 16);
  long cap = 2048 / p.nslabs;
  if (cap > EF_MAX_GX) cap = EF_MAX_GX;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  p.rpb = (rows + want - 1) / want;
  if (p.rpb < rpar) p.rpb = rpar;
  p.gx = cdiv(rows, p.rpb);
  return p;
}

}  // namespace spg

using namespace spg;

static inline int vec_of_e(int dtype) { return dtype == SPG_BF16 ? 8 : 4; }
#define EA_CHECK_C(what)                                                                                                                   \
  const int v = vec_of_e(dtype);                                                                                                           \
  SPG_REQUIRE(C % v == 0 && 256 % (C / v) == 0, what ": C=%d must be a multiple of %d with C/%d dividing 256", C, v, v)

constexpr int DW4_WGRAD_BLOCKS = 64;

extern "C" int spg_dwconv4(int dtype, const void* x, const float* const* w4, const int* dil4, void* y, int B, int H, int W, int C,
                           spg_stream_t stream) {
  EA_CHECK_C("dwconv4");
  Dw4 d;
  for (int i = 0; i < 4; ++i) { d.w[i] = w4[i]; d.dil[i] = dil4[i]; SPG_REQUIRE(w4[i] && dil4[i] >= 1, "dwconv4: branch %d", i); }
  long gx = ((long)B * H * W * (C / v) + 1023) / 1024;          // ~4 pixels per thread: the LDS weight stage is paid once per 4
  if (gx < 1) gx = 1;
  if (gx > 2048) gx = 2048;
  dim3 grid((int)gx, 4);
  const size_t lds = (size_t)9 * C * sizeof(float);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(dwconv4_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, d, (bf16_t*)y, B, H, W, C);
  else hipLaunchKernelGGL(dwconv4_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)x, d, (float*)y, B, H, W, C);
  return check_launch("dwconv4");
}

extern "C" int spg_dwconv4_dgrad(int dtype, const void* dy, const float* const* w4, const int* dil4, const float* gadd, void* dx, int B, int H,
                                 int W, int C, spg_stream_t stream) {
  EA_CHECK_C("dwconv4_dgrad");
  Dw4 d;
  for (int i = 0; i < 4; ++i) { d.w[i] = w4[i]; d.dil[i] = dil4[i]; }
  // (one item per thread: two -- the 4 x 9 x C weight stage paid once per two pixels -- measured slower, 32.8 -> 40.9 us)
  const int grid = ea_grid((long)B * H * W * (C / v));
  const size_t lds = (size_t)4 * 9 * C * sizeof(float);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(dwconv4_dgrad_kernel<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy, d, gadd, (bf16_t*)dx, B, H, W, C);
  else hipLaunchKernelGGL(dwconv4_dgrad_kernel<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float*)dy, d, gadd, (float*)dx, B, H, W, C);
  return check_launch("dwconv4_dgrad");
}

/* dw4: four f32 [C][9] gradients (+=).  red_ws: 4*64*9*C floats, red_counters: 4 zeroed words */
extern "C" int spg_dwconv4_wgrad(int dtype, const void* dy, const void* x, const int* dil4, float* const* dw4, int B, int H, int W, int C,
                                 float* red_ws, long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  EA_CHECK_C("dwconv4_wgrad");
  SPG_REQUIRE(9 * C <= 2048 && (9 * C) % 4 == 0, "dwconv4_wgrad: C=%d too wide", C);
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= 4L * DW4_WGRAD_BLOCKS * 9 * C, "dwconv4_wgrad: needs 4*64*9*C floats of scratch and 4 zeroed counters");
  Dw4 d;
  for (int i = 0; i < 4; ++i) { d.w[i] = nullptr; d.dil[i] = dil4[i]; }
  const long npix = (long)B * H * W;
  long ppb = cdiv(npix, DW4_WGRAD_BLOCKS);
  if (ppb < 64) ppb = 64;
  dim3 grid(cdiv(npix, ppb), 4);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(dwconv4_wgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, d, dw4[0], dw4[1], dw4[2], dw4[3], B, H, W, C, ppb, red_ws, red_counters_);
  else hipLaunchKernelGGL(dwconv4_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, d, dw4[0], dw4[1], dw4[2], dw4[3], B, H, W, C, ppb, red_ws, red_counters_);
  return check_launch("dwconv4_wgrad");
}

extern "C" int spg_easpp_fuse_bn(int dtype, const void* dcat, const float* scale_shift, const float* glob, const float* w, void* y, int B, long HW,
                                 int C, spg_stream_t stream) {
  const long total = (long)B * HW * C;
  SPG_REQUIRE(dcat && scale_shift && glob && w && y, "easpp_fuse_bn: null argument");
  if (dtype == SPG_BF16) hipLaunchKernelGGL(easpp_fuse_bn_kernel<bf16_t>, dim3(ea_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcat, scale_shift, glob, w, (bf16_t*)y, HW, C, total);
  else hipLaunchKernelGGL(easpp_fuse_bn_kernel<float>, dim3(ea_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dcat, scale_shift, glob, w, (float*)y, HW, C, total);
  return check_launch("easpp_fuse_bn");
}

extern "C" long spg_easpp_fuse_bn_bwd_workspace_floats(int dtype, int C) {
  const int v = vec_of_e(dtype), nch = 4 * C / v, nchs = nch < EF_SLAB_CHUNKS ? nch : EF_SLAB_CHUNKS, nslabs = cdiv(nch, nchs);
  long cap = 2048 / nslabs;
  if (cap > EF_MAX_GX) cap = EF_MAX_GX;
  return cap * nslabs * 3L * nchs * v;
}
extern "C" int spg_easpp_fuse_bn_bwd_counters(int dtype, int C) {
  const int v = vec_of_e(dtype), nch = 4 * C / v, nchs = nch < EF_SLAB_CHUNKS ? nch : EF_SLAB_CHUNKS;
  return cdiv(nch, nchs);
}

/* gamma4 / dgamma4 / dbeta4: HOST arrays of the four branch BatchNorms' parameter / gradient pointers (f32 [C] each); sums: f32 [12C] scratch */
extern "C" int spg_easpp_fuse_bn_bwd(int dtype, const void* dfu, const void* dcat, const float* w, const float* scale_shift, const float* mean_invstd,
                                     const float* const* gamma4, float* const* dgamma4, float* const* dbeta4, float* sums, void* ddcat, float* dw,
                                     int B, long HW, int C, float* red_ws, long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  const int v = vec_of_e(dtype);
  SPG_REQUIRE((4 * C) % v == 0 && 256 % (4 * C / v) == 0, "easpp_fuse_bn_bwd: 4C/%d must divide 256 (C=%d)", v, C);
  const long M = (long)B * HW;
  const EfPlan p = ef_plan(M, 4 * C / v);
  const long need = (long)p.gx * p.nslabs * 3 * p.nchs * v;
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= need, "easpp_fuse_bn_bwd: reduction workspace of %ld floats required, got %ld", need, red_ws_floats);
  Bn4 bn;
  for (int i = 0; i < 4; ++i) { bn.gamma[i] = gamma4[i]; bn.dgamma[i] = dgamma4[i]; bn.dbeta[i] = dbeta4[i]; }
  hipStream_t s = (hipStream_t)stream;
  const long total = M * (4 * C / v);
  // (a thread of the apply kernel first loads ~70 per-channel parameters: at one item per thread that prologue was most of the launch)
  const int apply_grid = ea_grid(total) > 1024 ? 1024 : ea_grid(total);
  if (dtype == SPG_BF16) {
    hipLaunchKernelGGL(easpp_fuse_bn_bwd_reduce_kernel<bf16_t>, dim3(p.gx, 1, p.nslabs), dim3(256), 0, s, (const bf16_t*)dfu, (const bf16_t*)dcat, w, scale_shift,
                       mean_invstd, sums, M, C, p.rpb, p.nchs, red_ws, red_counters_);
    hipLaunchKernelGGL(easpp_fuse_bn_bwd_apply_kernel<bf16_t>, dim3(apply_grid), dim3(256), 0, s, (const bf16_t*)dfu, (const bf16_t*)dcat, w, scale_shift,
                       mean_invstd, bn, sums, (bf16_t*)ddcat, dw, M, C, total);
  } else {
    hipLaunchKernelGGL(easpp_fuse_bn_bwd_reduce_kernel<float>, dim3(p.gx, 1, p.nslabs), dim3(256), 0, s, (const float*)dfu, (const float*)dcat, w, scale_shift,
                       mean_invstd, sums, M, C, p.rpb, p.nchs, red_ws, red_counters_);
    hipLaunchKernelGGL(easpp_fuse_bn_bwd_apply_kernel<float>, dim3(apply_grid), dim3(256), 0, s, (const float*)dfu, (const float*)dcat, w, scale_shift,
                       mean_invstd, bn, sums, (float*)ddcat, dw, M, C, total);
  }
  return check_launch("easpp_fuse_bn_bwd");
}

extern "C" int spg_easpp_global_fwd(const float* gsum, const float* Wg, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* gm, float* gl0, float* glob, float* scale_shift, float* mean_invstd, int B,
                                    int C, long HW, float eps, float momentum, int training, spg_stream_t stream) {
  SPG_REQUIRE(B >= 1 && B <= 64 && C >= 1 && C <= 512, "easpp_global_fwd: B=%d (1..64), C=%d (1..512)", B, C);
  SPG_REQUIRE(training || (running_mean && running_var), "easpp_global_fwd: eval mode needs running statistics");
  SPG_REQUIRE(!training || B >= 2, "easpp_global_fwd: train-mode BatchNorm over B values needs B >= 2 (Expected more than 1 value per channel)");
  SPG_REQUIRE((long)B * C <= 8192, "easpp_global_fwd: B * C = %ld exceeds the 64 KiB of LDS the kernel stages (8192 values)", (long)B * C);
  // the weight matrix is staged in LDS when it fits beside the two [B][C] arrays (C = 128: 64 KiB + 8 KiB at batch 8)
  const size_t lds_w = (size_t)C * (C + 4) * sizeof(float), lds_base = (size_t)2 * B * C * sizeof(float);
  const int stage_w = (C % 4 == 0 && ((uintptr_t)Wg & 15) == 0 && lds_base + lds_w <= 128 * 1024) ? 1 : 0;
  const size_t lds = lds_base + (stage_w ? lds_w : 0);
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      SPG_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&easpp_global_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024) == hipSuccess,
                  "easpp_global_fwd: cannot raise the dynamic LDS limit");
      raised = true;
    }
  }
  hipLaunchKernelGGL(easpp_global_fwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, gsum, Wg, gamma, beta, running_mean,
                     running_var, num_batches_tracked, gm, gl0, glob, scale_shift, mean_invstd, B, C, 1.f / (float)HW, eps, momentum, training, stage_w);
  return check_launch("easpp_global_fwd");
}

extern "C" int spg_easpp_global_bwd(const float* S, const float* glob, const float* gl0, const float* gm, const float* wf, const float* Wg,
                                    const float* gamma, const float* mean_invstd, float* dwf, float* dWg, float* dgamma, float* dbeta, float* gadd,
                                    int B, int C, long HW, int training, spg_stream_t stream) {
  SPG_REQUIRE(B >= 1 && B <= 64 && C >= 1 && C <= 512, "easpp_global_bwd: B=%d (1..64), C=%d (1..512)", B, C);
  SPG_REQUIRE((long)B * C <= 8192, "easpp_global_bwd: B * C = %ld exceeds the 64 KiB of LDS the kernel stages (8192 values)", (long)B * C);
  hipLaunchKernelGGL(easpp_global_bwd_kernel, dim3(1), dim3(1024), (size_t)2 * B * C * sizeof(float), (hipStream_t)stream, S, glob, gl0, gm, wf, Wg, gamma,
                     mean_invstd, dwf, dWg, dgamma, dbeta, gadd, B, C, 1.f / (float)HW, training);
  return check_launch("easpp_global_bwd");
}
