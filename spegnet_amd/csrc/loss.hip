// Fused CODLoss for fixed-size ground truth (reference utils/loss_functions.py:114-295 + the resize loop of
// engine/trainer.py:358-383), forward sums and analytic gradients, no autograd graph and no per-sample launches.
//
//   wmap      : w = 1 + bw*(|Laplace3x3(m)| + |avgpool31(m) - m|) per pixel, per-image sum(m), sum(w), sum(edge_gt)
//   seg_reduce: per scale and image  A = sum w*bce(z,m;pw), I = sum s*m*w, U = sum (s+m)*w,  z = bilinear(pred -> SxS)
//   edge_reduce: per image            F = sum focal(z,t;pw), I = sum s*t, P = sum s
//   finalize  : loss = sum_i w_i * mean_b(bce_w*A/W + iou_w*(1-(I+1)/(U-I+1))) + edge_w * mean_b(F/HW + 1-(2I+1)/(P+T+1))
//   seg_grad / edge_grad: d loss / d pred at the PREDICTION's resolution = bilinear-adjoint gather of the per-pixel
//   gradient (each low-res logit sums over the full-res pixels whose interpolation touches it), scaled by the upstream
//   gradient read from device memory (no host sync).
// All HBM-trivial (a few MB); the point is launch count and staying inside the hipGraph.
#include <type_traits>
#include "common.h"

namespace spg {

// blocks per image of the per-image reductions; the one-map and all-maps entry points use the same count, so their sums are bit-identical.
// (256 blocks -- 2-3 pixels per thread instead of 9 -- made the all-maps launch SLOWER, 39 -> 102 us: 32 last arrivers each finish 256 partials.)
constexpr int LOSS_RED_BLOCKS = 64;

__device__ __forceinline__ void bil_src_l(int dst, int in, int out, int& i0, int& i1, float& lam) {
  float src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
  src = fmaxf(src, 0.f);
  i0 = min((int)src, in - 1);
  i1 = min(i0 + 1, in - 1);
  lam = src - (float)i0;
}
template <typename T>
__device__ __forceinline__ float bil_at(const T* __restrict__ p, int h, int w, int S, int Y, int X) {
  if (h == S && w == S) return ST<T>::ld(p + (long)Y * w + X);
  int y0, y1, x0, x1; float ly, lx;
  bil_src_l(Y, h, S, y0, y1, ly);
  bil_src_l(X, w, S, x0, x1, lx);
  const float a = ST<T>::ld(p + (long)y0 * w + x0), b = ST<T>::ld(p + (long)y0 * w + x1);
  const float c = ST<T>::ld(p + (long)y1 * w + x0), d = ST<T>::ld(p + (long)y1 * w + x1);
  return (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
}
// bil_at in two halves: the four taps requested now, interpolated later (the reductions below request three pixels' operands per trip before
// using any -- one pixel per trip was a memory round trip per pixel, nine in a row at 384 x 384)
struct BilTaps { float a, b, c, d, ly, lx; };
template <typename T, bool IDENT>
__device__ __forceinline__ BilTaps bil_request(const T* __restrict__ p, int h, int w, int S, int Y, int X) {
  BilTaps t;
  if constexpr (IDENT) { t.a = ST<T>::ld(p + (long)Y * w + X); t.b = t.c = t.d = 0.f; t.ly = t.lx = 0.f; return t; }
  int y0, y1, x0, x1;
  bil_src_l(Y, h, S, y0, y1, t.ly);
  bil_src_l(X, w, S, x0, x1, t.lx);
  t.a = ST<T>::ld(p + (long)y0 * w + x0); t.b = ST<T>::ld(p + (long)y0 * w + x1);
  t.c = ST<T>::ld(p + (long)y1 * w + x0); t.d = ST<T>::ld(p + (long)y1 * w + x1);
  return t;
}
template <bool IDENT>
__device__ __forceinline__ float bil_value(const BilTaps& t) {
  if constexpr (IDENT) return t.a;
  return (1.f - t.ly) * ((1.f - t.lx) * t.a + t.lx * t.b) + t.ly * ((1.f - t.lx) * t.c + t.lx * t.d);
}
__device__ __forceinline__ float pos_weight(float npos, float total) {
  return fminf(fmaxf((total - npos) / (npos + 1e-7f), 0.1f), 10.f);
}

// stats[b] = {sum m, sum w, sum edge_gt, 0}
__global__ __launch_bounds__(256) void loss_wmap_kernel(const float* __restrict__ mask, const float* __restrict__ egt,
                                                        float* __restrict__ wmap, float* __restrict__ stats, int S, float bw,
                                                        float* __restrict__ part, unsigned* __restrict__ counters) {
  __shared__ float tile[62][65];
  __shared__ float hs[62][33];
  __shared__ float red[4];
  const int b = blockIdx.z, ty0 = blockIdx.y * 32, tx0 = blockIdx.x * 32;
  const float* m = mask + (long)b * S * S;
  // (the 62 x 62 halo tile, five elements per thread and trip with their loads requested together: `cond ? load : 0` per element was one
  // memory round trip for each of a thread's 16 elements)
  for (int i0 = threadIdx.x; i0 < 62 * 62; i0 += 256 * 5) {
    float v[5];
    bool in[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = i0 + u * 256;
      const int r = i / 62, c = i - r * 62;
      const int y = ty0 + r - 15, x = tx0 + c - 15;
      in[u] = i < 62 * 62 && (unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S;
      v[u] = m[in[u] ? (long)y * S + x : 0];
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = i0 + u * 256;
      if (i < 62 * 62) { const int r = i / 62, c = i - r * 62; tile[r][c] = in[u] ? v[u] : 0.f; }
    }
  }
  __syncthreads();
  // 31-tap box sums as running sums: a thread owns 8 consecutive outputs of a row (31 + 2 x 7 LDS reads instead of 8 x 31), then 4
  // consecutive outputs of a column.  (Binary masks make every partial sum an exact integer; for soft masks the order of the adds differs
  // from a tap-by-tap sum in the last bit.)
  if (threadIdx.x < 62 * 4) {
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 8;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 31; ++d) s += tile[r][c0 + d];
    hs[r][c0] = s;
#pragma unroll
    for (int j = 1; j < 8; ++j) { s += tile[r][c0 + j + 30] - tile[r][c0 + j - 1]; hs[r][c0 + j] = s; }
  }
  __syncthreads();
  float sm = 0.f, sw = 0.f, se = 0.f;
  {
    const int c = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 31; ++d) s += hs[r0 + d][c];
    float ev[4];                               // (the four edge-map values of this thread's outputs, requested together)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = ty0 + r0 + j, x = tx0 + c;
      ev[j] = egt[(y < S && x < S) ? ((long)b * S + y) * S + x : (long)b * S * S];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + j;
      if (j) s += hs[r + 30][c] - hs[r - 1][c];
      const int y = ty0 + r, x = tx0 + c;
      if (y < S && x < S) {
        const float mv = tile[r + 15][c + 15];
        float nb = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) nb += tile[r + 15 + dy][c + 15 + dx];
        const float lap = fabsf(9.f * mv - nb);                       // 8*m - sum(8 neighbours)
        const float w = 1.f + bw * (lap + fabsf(s * (1.f / 961.f) - mv));
        wmap[((long)b * S + y) * S + x] = w;
        sm += mv; sw += w;
        se += ev[j];
      }
    }
  }
  sm = block_sum<256>(sm, red);
  sw = block_sum<256>(sw, red);
  se = block_sum<256>(se, red);
  // deterministic finish: one partial triple per tile, the image's last tile adds them in tile order
  __shared__ unsigned s_last;
  const int nblk = gridDim.x * gridDim.y, bl = blockIdx.y * gridDim.x + blockIdx.x;
  float* pp = part + ((long)b * nblk + bl) * 4;
  if (threadIdx.x == 0) { st_part(pp, sm); st_part(pp + 1, sw); st_part(pp + 2, se); }
  if (!arrive_last(counters + b, (unsigned)nblk, &s_last)) return;
  finish_partials<256>(part + (long)b * nblk * 4, nblk, 4, 3, stats + b * 4, 0, &hs[0][0]);
}

// sums[b] = {A, I, U}.  (b, bx, gx) = image, block and blocks per image: the one-map and the all-maps kernels run the same body on the
// same (gx, B) block grid, so their partials and the fixed-order finish are bit-identical.
template <typename T>
__device__ __forceinline__ void seg_reduce_body(const T* __restrict__ pred, const float* __restrict__ mask,
                                                const float* __restrict__ wmap, const float* __restrict__ stats,
                                                float* __restrict__ sums, int S, int h, int w,
                                                float* __restrict__ part, unsigned* __restrict__ counters, int b, int bx, int gx) {
  __shared__ float red[4];
  __shared__ float scratch[256];
  __shared__ unsigned s_last;
  const long HW = (long)S * S;
  const float pw = pos_weight(stats[b * 4 + 0], (float)HW);
  const T* p = pred + (long)b * h * w;
  float A = 0.f, I = 0.f, U = 0.f;
  const long stride = (long)gx * 256;
  auto walk = [&](auto ident_c) __attribute__((always_inline)) {
    constexpr bool IDENT = decltype(ident_c)::value;
    for (long i0 = bx * 256L + threadIdx.x; i0 < HW; i0 += 3 * stride) {
      BilTaps tp[3];
      float mv[3], wvv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const long i = i0 + u * stride < HW ? i0 + u * stride : i0;
        const int Y = (int)(i / S), X = (int)(i - (long)Y * S);
        tp[u] = bil_request<T, IDENT>(p, h, w, S, Y, X);
        mv[u] = mask[b * HW + i]; wvv[u] = wmap[b * HW + i];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (i0 + u * stride < HW) {
          const float z = bil_value<IDENT>(tp[u]);
          const float m = mv[u], wv = wvv[u];
          const float lw = 1.f + (pw - 1.f) * m;
          const float bce = (1.f - m) * z + lw * (log1pf(__expf(-fabsf(z))) + fmaxf(-z, 0.f));
          const float s = sigmoid_f(z);
          A += wv * bce; I += s * m * wv; U += (s + m) * wv;
        }
      }
    }
  };
  if (h == S && w == S) walk(std::true_type{}); else walk(std::false_type{});
  A = block_sum<256>(A, red); I = block_sum<256>(I, red); U = block_sum<256>(U, red);
  float* pp = part + ((long)b * gx + bx) * 4;
  if (threadIdx.x == 0) { st_part(pp, A); st_part(pp + 1, I); st_part(pp + 2, U); }
  if (!arrive_last(counters + b, gx, &s_last)) return;
  finish_partials<256>(part + (long)b * gx * 4, gx, 4, 3, sums + b * 3, 0, scratch);
}
template <typename T>
__global__ __launch_bounds__(256) void loss_seg_reduce_kernel(const T* __restrict__ pred, const float* __restrict__ mask,
                                                              const float* __restrict__ wmap, const float* __restrict__ stats,
                                                              float* __restrict__ sums, int S, int h, int w,
                                                              float* __restrict__ part, unsigned* __restrict__ counters) {
  seg_reduce_body<T>(pred, mask, wmap, stats, sums, S, h, w, part, counters, blockIdx.y, blockIdx.x, gridDim.x);
}

// sums[b] = {F, I, P}
template <typename T>
__device__ __forceinline__ void edge_reduce_body(const T* __restrict__ pred, const float* __restrict__ egt,
                                                 const float* __restrict__ stats, float* __restrict__ sums, int S,
                                                 int h, int w, float alpha, float gamma,
                                                 float* __restrict__ part, unsigned* __restrict__ counters, int b, int bx, int gx) {
  __shared__ float red[4];
  __shared__ float scratch[256];
  __shared__ unsigned s_last;
  const long HW = (long)S * S;
  const float pw = pos_weight(stats[b * 4 + 2], (float)HW);
  const T* p = pred + (long)b * h * w;
  float Fs = 0.f, I = 0.f, P = 0.f;
  const long stride = (long)gx * 256;
  auto walk = [&](auto ident_c) __attribute__((always_inline)) {
    constexpr bool IDENT = decltype(ident_c)::value;
    for (long i0 = bx * 256L + threadIdx.x; i0 < HW; i0 += 3 * stride) {
      BilTaps tp[3];
      float tv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const long i = i0 + u * stride < HW ? i0 + u * stride : i0;
        const int Y = (int)(i / S), X = (int)(i - (long)Y * S);
        tp[u] = bil_request<T, IDENT>(p, h, w, S, Y, X);
        tv[u] = egt[b * HW + i];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (i0 + u * stride < HW) {
          const float z = bil_value<IDENT>(tp[u]);
          const float t = tv[u];
          const float s = sigmoid_f(z);
          const float pt = t * s + (1.f - t) * (1.f - s);
          Fs += -pw * alpha * __powf(1.f - pt, gamma) * __logf(fmaxf(pt, 1e-7f));
          I += s * t; P += s;
        }
      }
    }
  };
  if (h == S && w == S) walk(std::true_type{}); else walk(std::false_type{});
  Fs = block_sum<256>(Fs, red); I = block_sum<256>(I, red); P = block_sum<256>(P, red);
  float* pp = part + ((long)b * gx + bx) * 4;
  if (threadIdx.x == 0) { st_part(pp, Fs); st_part(pp + 1, I); st_part(pp + 2, P); }
  if (!arrive_last(counters + b, gx, &s_last)) return;
  finish_partials<256>(part + (long)b * gx * 4, gx, 4, 3, sums + b * 3, 0, scratch);
}
template <typename T>
__global__ __launch_bounds__(256) void loss_edge_reduce_kernel(const T* __restrict__ pred, const float* __restrict__ egt,
                                                               const float* __restrict__ stats, float* __restrict__ sums, int S,
                                                               int h, int w, float alpha, float gamma,
                                                               float* __restrict__ part, unsigned* __restrict__ counters) {
  edge_reduce_body<T>(pred, egt, stats, sums, S, h, w, alpha, gamma, part, counters, blockIdx.y, blockIdx.x, gridDim.x);
}

// The four maps of the loss (three segmentation scales + the edge map) in ONE launch: blockIdx.z = map.  Each was a (blocks) x B launch
// of ~15-20 us that the chip spends mostly waiting on (1.2 M pixels); side by side they take the time of one.
struct LossMaps {
  const void* pred[4];   // [0..2] segmentation logits (coarse to fine), [3] edge logits
  void* dpred[4];        // gradients (loss_grad_all only)
  int h[4], w[4];
  float coef[4];         // gradient scale per map: scale weight / B (segmentation), edge weight / B
};
template <typename T>
__global__ __launch_bounds__(256) void loss_reduce_all_kernel(LossMaps ms, const float* __restrict__ mask, const float* __restrict__ egt,
                                                              const float* __restrict__ wmap, const float* __restrict__ stats,
                                                              float* __restrict__ seg_sums, float* __restrict__ edge_sums, int B, int S,
                                                              float alpha, float gamma, float* __restrict__ part, unsigned* __restrict__ counters) {
  const int z = blockIdx.z;
  float* pz = part + (long)z * B * gridDim.x * 4;
  unsigned* cz = counters + z * B;
  const T* pred = (const T*)(z == 0 ? ms.pred[0] : z == 1 ? ms.pred[1] : z == 2 ? ms.pred[2] : ms.pred[3]);
  const int h = z == 0 ? ms.h[0] : z == 1 ? ms.h[1] : z == 2 ? ms.h[2] : ms.h[3];
  const int w = z == 0 ? ms.w[0] : z == 1 ? ms.w[1] : z == 2 ? ms.w[2] : ms.w[3];
  if (z < 3) seg_reduce_body<T>(pred, mask, wmap, stats, seg_sums + (long)z * B * 3, S, h, w, pz, cz, blockIdx.y, blockIdx.x, gridDim.x);
  else edge_reduce_body<T>(pred, egt, stats, edge_sums, S, h, w, alpha, gamma, pz, cz, blockIdx.y, blockIdx.x, gridDim.x);
}

struct LossCfg {
  float sw[3];
  float bce_w, iou_w, edge_w, alpha, gamma;
};

// out = {loss, seg_loss, edge_loss}
__global__ void loss_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ seg_sums /*[3][B][3]*/,
                                     const float* __restrict__ edge_sums, float* __restrict__ out, int B, long HW, LossCfg c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float seg = 0.f, edge = 0.f;
  for (int b = 0; b < B; ++b) {
    const float W = stats[b * 4 + 1];
    for (int i = 0; i < 3; ++i) {
      const float* s = seg_sums + ((long)i * B + b) * 3;
      seg += c.sw[i] * (c.bce_w * s[0] / W + c.iou_w * (1.f - (s[1] + 1.f) / (s[2] - s[1] + 1.f)));
    }
    const float* e = edge_sums + b * 3;
    edge += e[0] / (float)HW + 1.f - (2.f * e[1] + 1.f) / (e[2] + stats[b * 4 + 2] + 1.f);
  }
  seg /= (float)B; edge /= (float)B;
  out[0] = seg + c.edge_w * edge; out[1] = seg; out[2] = edge;
}

// gradient w.r.t. the low-res prediction: gather over the full-res pixels whose bilinear taps include (yl, xl).
// One 64-lane wave per low-res pixel: the lanes split the (3*sy) x (3*sx) candidate window and wave-reduce, so the 48x48
// edge map (576 candidates per pixel) still launches 18k waves instead of 18k threads.
template <typename T, bool EDGE>
__global__ __launch_bounds__(256) void loss_grad_kernel(const T* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ wmap, const float* __restrict__ stats,
                                                        const float* __restrict__ sums, const float* __restrict__ go,
                                                        T* __restrict__ dpred, int B, int S, int h, int w, float coef, float bce_w,
                                                        float iou_w, float alpha, float gamma) {
  const long total = (long)B * h * w;
  const long HW = (long)S * S;
  const int sy = S / h, sx = S / w;
  const float g0 = go ? go[0] : 1.f;
  // identity scale (prediction already at the target resolution): one THREAD per pixel; otherwise one wave per pixel
  const bool ident = (h == S && w == S);
  const int lane = ident ? 0 : (threadIdx.x & 63);
  const int lstep = ident ? 1 : 64;
  const long wave0 = ident ? (blockIdx.x * 256L + threadIdx.x) : ((blockIdx.x * 256L + threadIdx.x) >> 6);
  const long nwaves = ident ? ((long)gridDim.x * 256) : (((long)gridDim.x * 256) >> 6);
  for (long i = wave0; i < total; i += nwaves) {
    const int xl = (int)(i % w);
    const int yl = (int)((i / w) % h);
    const int b = (int)(i / ((long)w * h));
    const T* p = pred + (long)b * h * w;
    float pw, k1 = 0.f, k2 = 0.f, invW = 0.f;
    if constexpr (EDGE) {
      pw = pos_weight(stats[b * 4 + 2], (float)HW);
      const float I = sums[b * 3 + 1], Ud = sums[b * 3 + 2] + stats[b * 4 + 2] + 1.f;   // dice = 1 - (2I+1)/Ud
      k1 = -2.f / Ud;                    // d dice / d I
      k2 = (2.f * I + 1.f) / (Ud * Ud);  // d dice / d P
    } else {
      pw = pos_weight(stats[b * 4 + 0], (float)HW);
      invW = 1.f / stats[b * 4 + 1];
      const float I = sums[b * 3 + 1], D = sums[b * 3 + 2] - I + 1.f;   // wiou = 1 - (I+1)/D, D = U - I + 1
      k1 = -(D + (I + 1.f)) / (D * D);   // d wiou / d I  (dD/dI = -1)
      k2 = (I + 1.f) / (D * D);          // d wiou / d U
    }
    const int Y0 = max(0, (yl - 1) * sy), Y1 = min(S - 1, (yl + 2) * sy);
    const int X0 = max(0, (xl - 1) * sx), X1 = min(S - 1, (xl + 2) * sx);
    const int nx = X1 - X0 + 1, ncand = (Y1 - Y0 + 1) * nx;
    float acc = 0.f;
    for (int c = lane; c < ncand; c += lstep) {
      const int Y = Y0 + c / nx, X = X0 + c % nx;
      float wy, wx;
      if (h == S) wy = (Y == yl) ? 1.f : 0.f;
      else { int y0, y1; float ly; bil_src_l(Y, h, S, y0, y1, ly); wy = (y0 == yl ? 1.f - ly : 0.f) + (y1 == yl ? ly : 0.f); }
      if (w == S) wx = (X == xl) ? 1.f : 0.f;
      else { int x0, x1; float lx; bil_src_l(X, w, S, x0, x1, lx); wx = (x0 == xl ? 1.f - lx : 0.f) + (x1 == xl ? lx : 0.f); }
      const float wgt = wy * wx;
      if (wgt == 0.f) continue;
      const float z = bil_at<T>(p, h, w, S, Y, X);
      const float t = tgt[b * HW + (long)Y * S + X];
      const float s = sigmoid_f(z), ds = s * (1.f - s);
      float g;
      if constexpr (EDGE) {
        const float pt = t * s + (1.f - t) * (1.f - s);
        const float ptc = fmaxf(pt, 1e-7f);
        const float om = 1.f - pt;
        // f = -pw*alpha*om^gamma*log(ptc);  df/dpt = -pw*alpha*( -gamma*om^(gamma-1)*log(ptc) + om^gamma/ptc*[pt>1e-7] )
        const float dfdpt = -pw * alpha * (-gamma * __powf(om, gamma - 1.f) * __logf(ptc) + (pt > 1e-7f ? __powf(om, gamma) / ptc : 0.f));
        const float dptdz = (2.f * t - 1.f) * ds;
        g = dfdpt * dptdz / (float)HW + (k1 * t + k2) * ds;
      } else {
        const float wv = wmap[b * HW + (long)Y * S + X];
        const float lw = 1.f + (pw - 1.f) * t;
        const float dbce = (1.f - t) - lw * (1.f - s);
        g = bce_w * wv * dbce * invW + iou_w * (k1 * t + k2) * ds * wv;
      }
      acc += wgt * g;
    }
    if (!ident) acc = wave_sum(acc);
    if (lane == 0) ST<T>::st(dpred + i, acc * coef * g0);
  }
}


// ---- the same gradient in two passes (used whenever the prediction is coarser than the target): the kernel above evaluates the loss
// derivative at every full-res pixel once PER low-res logit whose window covers it -- (3 sy)(3 sx) = 144 / 36 / 576 evaluations per
// logit, ~10.6 M sigmoid + focal / BCE evaluations per map at batch 8 against 1.18 M pixels.  Pass 1 writes dL/dz of every full-res pixel
// once; pass 2 is the adjoint of the bilinear up-sampling (weights + one load per candidate).
template <typename T, bool EDGE>
__device__ __forceinline__ float loss_dz_at(const T* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ wmap,
                                            const float* __restrict__ stats, const float* __restrict__ sums, long i, int S, int h, int w,
                                            float bce_w, float iou_w, float alpha, float gamma) {
  const long HW = (long)S * S;
  const int b = (int)(i / HW);
  const long pix = i - (long)b * HW;
  const int Y = (int)(pix / S), X = (int)(pix - (long)Y * S);
  const float z = bil_at<T>(pred + (long)b * h * w, h, w, S, Y, X);
  const float t = tgt[i];
  const float s = sigmoid_f(z), ds = s * (1.f - s);
  float g;
  if constexpr (EDGE) {
    const float pw = pos_weight(stats[b * 4 + 2], (float)HW);
    const float I = sums[b * 3 + 1], Ud = sums[b * 3 + 2] + stats[b * 4 + 2] + 1.f;
    const float k1 = -2.f / Ud, k2 = (2.f * I + 1.f) / (Ud * Ud);
    const float pt = t * s + (1.f - t) * (1.f - s);
    const float ptc = fmaxf(pt, 1e-7f);
    const float om = 1.f - pt;
    const float dfdpt = -pw * alpha * (-gamma * __powf(om, gamma - 1.f) * __logf(ptc) + (pt > 1e-7f ? __powf(om, gamma) / ptc : 0.f));
    const float dptdz = (2.f * t - 1.f) * ds;
    g = dfdpt * dptdz / (float)HW + (k1 * t + k2) * ds;
  } else {
    const float pw = pos_weight(stats[b * 4 + 0], (float)HW);
    const float invW = 1.f / stats[b * 4 + 1];
    const float I = sums[b * 3 + 1], D = sums[b * 3 + 2] - I + 1.f;
    const float k1 = -(D + (I + 1.f)) / (D * D), k2 = (I + 1.f) / (D * D);
    const float wv = wmap[i];
    const float lw = 1.f + (pw - 1.f) * t;
    const float dbce = (1.f - t) - lw * (1.f - s);
    g = bce_w * wv * dbce * invW + iou_w * (k1 * t + k2) * ds * wv;
  }
  return g;
}
template <typename T, bool EDGE>
__global__ __launch_bounds__(256) void loss_dz_kernel(const T* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ wmap,
                                                      const float* __restrict__ stats, const float* __restrict__ sums, float* __restrict__ dz,
                                                      int B, int S, int h, int w, float bce_w, float iou_w, float alpha, float gamma) {
  const long total = (long)B * S * S;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    dz[i] = loss_dz_at<T, EDGE>(pred, tgt, wmap, stats, sums, i, S, h, w, bce_w, iou_w, alpha, gamma);
}
// all four maps, blockIdx.y = map.  A map already at the target's resolution needs no second pass: its gradient is written here.
template <typename T>
__global__ __launch_bounds__(256) void loss_dz_all_kernel(LossMaps ms, const float* __restrict__ mask, const float* __restrict__ egt,
                                                          const float* __restrict__ wmap, const float* __restrict__ stats,
                                                          const float* __restrict__ seg_sums, const float* __restrict__ edge_sums,
                                                          const float* __restrict__ go, float* __restrict__ dz, int B, int S, float bce_w,
                                                          float iou_w, float alpha, float gamma) {
  const int z = blockIdx.y;
  const long total = (long)B * S * S;
  const T* pred = (const T*)(z == 0 ? ms.pred[0] : z == 1 ? ms.pred[1] : z == 2 ? ms.pred[2] : ms.pred[3]);
  T* dpred = (T*)(z == 0 ? ms.dpred[0] : z == 1 ? ms.dpred[1] : z == 2 ? ms.dpred[2] : ms.dpred[3]);
  const int h = z == 0 ? ms.h[0] : z == 1 ? ms.h[1] : z == 2 ? ms.h[2] : ms.h[3];
  const int w = z == 0 ? ms.w[0] : z == 1 ? ms.w[1] : z == 2 ? ms.w[2] : ms.w[3];
  const float coef = z == 0 ? ms.coef[0] : z == 1 ? ms.coef[1] : z == 2 ? ms.coef[2] : ms.coef[3];
  const bool ident = (h == S && w == S);
  const float g0 = go ? go[0] : 1.f;
  float* dzz = dz + (long)z * total;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float g = z < 3 ? loss_dz_at<T, false>(pred, mask, wmap, stats, seg_sums + (long)z * B * 3, i, S, h, w, bce_w, iou_w, alpha, gamma)
                          : loss_dz_at<T, true>(pred, egt, wmap, stats, edge_sums, i, S, h, w, bce_w, iou_w, alpha, gamma);
    if (ident) ST<T>::st(dpred + i, g * coef * g0);
    else dzz[i] = g;
  }
}
// pass 2: a group of L lanes per low-res logit over its (2 sy) x (2 sx) candidate window ((3 sy) x (3 sx) for an odd scale; candidates
// outside its taps weigh 0).  L is sized to the window -- 4 lanes for the x2 scale's 16 candidates, 16 for x4's 64, a whole wave for x8's
// 256: with a wave per logit the x2 scale (295 k logits at batch 8) ran 16 of 64 lanes for one load each, 38 us per launch.
template <typename T, int L>
__device__ __forceinline__ void loss_gather_body(const float* __restrict__ dz, float g0, T* __restrict__ dpred,
                                                 int B, int S, int h, int w, float coef, long blk, long nblk) {
  const long total = (long)B * h * w;
  const long HW = (long)S * S;
  const int sy = S / h, sx = S / w;
  const int sub = threadIdx.x & (L - 1);
  const long grp0 = (blk * 256L + threadIdx.x) / L, ngrp = (nblk * 256) / L;
  const long iters = (total + ngrp - 1) / ngrp;             // (every lane of a wave runs the same number of rounds: the shuffles need them all)
  for (long it = 0; it < iters; ++it) {
    const long i = grp0 + it * ngrp;
    const bool live = i < total;
    const long ii = live ? i : 0;
    const int xl = (int)(ii % w);
    const int yl = (int)((ii / w) % h);
    const int b = (int)(ii / ((long)w * h));
    // full-res pixels whose taps include (yl, xl): source coordinate (Y + 0.5) / sy - 0.5 in [yl - 1, yl + 1), i.e. for an even scale
    // exactly the 2 sy rows [yl sy - sy/2, yl sy + 3 sy/2 - 1] (clamped: the border clamps of the interpolation fall inside); an odd
    // scale keeps the conservative (3 sy)-row window -- candidates outside the taps weigh 0 either way
    const int Y0 = (sy & 1) ? max(0, (yl - 1) * sy) : max(0, yl * sy - sy / 2), Y1 = (sy & 1) ? min(S - 1, (yl + 2) * sy) : min(S - 1, yl * sy + 3 * sy / 2 - 1);
    const int X0 = (sx & 1) ? max(0, (xl - 1) * sx) : max(0, xl * sx - sx / 2), X1 = (sx & 1) ? min(S - 1, (xl + 2) * sx) : min(S - 1, xl * sx + 3 * sx / 2 - 1);
    const int nx = X1 - X0 + 1, ncand = live ? (Y1 - Y0 + 1) * nx : 0;
    float acc = 0.f;
    for (int c = sub; c < ncand; c += L) {
      const int Y = Y0 + c / nx, X = X0 + c % nx;
      float wy, wx;
      { int y0, y1; float ly; bil_src_l(Y, h, S, y0, y1, ly); wy = (y0 == yl ? 1.f - ly : 0.f) + (y1 == yl ? ly : 0.f); }
      { int x0, x1; float lx; bil_src_l(X, w, S, x0, x1, lx); wx = (x0 == xl ? 1.f - lx : 0.f) + (x1 == xl ? lx : 0.f); }
      const float wgt = wy * wx;
      if (wgt != 0.f) acc += wgt * dz[b * HW + (long)Y * S + X];     // (all four candidates' loads in flight instead: 37.8 -> 38.9 us, the weights' arithmetic bounds this pass)
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && live) ST<T>::st(dpred + i, acc * coef * g0);
  }
}
template <typename T, int L>
__global__ __launch_bounds__(256) void loss_gather_kernel(const float* __restrict__ dz, const float* __restrict__ go, T* __restrict__ dpred,
                                                          int B, int S, int h, int w, float coef) {
  loss_gather_body<T, L>(dz, go ? go[0] : 1.f, dpred, B, S, h, w, coef, blockIdx.x, gridDim.x);
}
__host__ __device__ inline int loss_gather_lanes(int S, int h, int w) {   // lanes per logit: a quarter of the even-scale window's candidates each
  const int win = (S / h) * (S / w) * (((S / h) & 1) ? 9 : 4);
  return win <= 16 ? 4 : (win <= 64 ? 16 : 64);
}
template <typename T>
__global__ __launch_bounds__(256) void loss_gather_all_kernel(LossMaps ms, const float* __restrict__ dz, const float* __restrict__ go, int B, int S) {
  const int z = blockIdx.y;
  const int h = z == 0 ? ms.h[0] : z == 1 ? ms.h[1] : z == 2 ? ms.h[2] : ms.h[3];
  const int w = z == 0 ? ms.w[0] : z == 1 ? ms.w[1] : z == 2 ? ms.w[2] : ms.w[3];
  if (h == S && w == S) return;                    // written by the first pass
  T* dpred = (T*)(z == 0 ? ms.dpred[0] : z == 1 ? ms.dpred[1] : z == 2 ? ms.dpred[2] : ms.dpred[3]);
  const float coef = z == 0 ? ms.coef[0] : z == 1 ? ms.coef[1] : z == 2 ? ms.coef[2] : ms.coef[3];
  const float g0 = go ? go[0] : 1.f;
  const float* dzz = dz + (long)z * B * S * S;
  const int L = loss_gather_lanes(S, h, w);
  if (L == 4) loss_gather_body<T, 4>(dzz, g0, dpred, B, S, h, w, coef, blockIdx.x, gridDim.x);
  else if (L == 16) loss_gather_body<T, 16>(dzz, g0, dpred, B, S, h, w, coef, blockIdx.x, gridDim.x);
  else loss_gather_body<T, 64>(dzz, g0, dpred, B, S, h, w, coef, blockIdx.x, gridDim.x);
}

}  // namespace spg

using namespace spg;

/* partial floats the two reductions need (wmap tiles / reduce blocks, 4 floats each); both also need B zeroed counters */
extern "C" long spg_loss_workspace_floats(int B, int S) {
  const long tiles = (long)cdiv(S, 32) * cdiv(S, 32);
  return (long)B * 4 * (tiles > LOSS_RED_BLOCKS ? tiles : LOSS_RED_BLOCKS);
}

extern "C" int spg_loss_weight_map(const float* mask, const float* edge_gt, float* wmap, float* stats, int B, int S,
                                   float boundary_weight, float* red_ws, long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  SPG_REQUIRE(B > 0 && S > 0, "loss_weight_map: empty");
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= spg_loss_workspace_floats(B, S), "loss_weight_map: reduction workspace too small");
  hipLaunchKernelGGL(loss_wmap_kernel, dim3(cdiv(S, 32), cdiv(S, 32), B), dim3(256), 0, (hipStream_t)stream, mask, edge_gt, wmap, stats, S, boundary_weight,
                     red_ws, red_counters_);
  return check_launch("loss_weight_map");
}

extern "C" int spg_loss_reduce(int dtype, const void* pred, const float* target, const float* wmap, const float* stats,
                               float* sums, int B, int S, int h, int w, int edge, float alpha, float gamma, float* red_ws,
                               long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  SPG_REQUIRE(S % h == 0 && S % w == 0, "loss_reduce: target size %d must be a multiple of the prediction size %dx%d", S, h, w);
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= (long)B * LOSS_RED_BLOCKS * 4, "loss_reduce: reduction workspace too small");
  long per = ((long)S * S + 255) / 256;
  int gx = (int)(per < LOSS_RED_BLOCKS ? per : LOSS_RED_BLOCKS);
  dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
  if (edge) {
    if (dtype == SPG_BF16) hipLaunchKernelGGL(loss_edge_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)pred, target, stats, sums, S, h, w, alpha, gamma, red_ws, red_counters_);
    else hipLaunchKernelGGL(loss_edge_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)pred, target, stats, sums, S, h, w, alpha, gamma, red_ws, red_counters_);
  } else {
    if (dtype == SPG_BF16) hipLaunchKernelGGL(loss_seg_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)pred, target, wmap, stats, sums, S, h, w, red_ws, red_counters_);
    else hipLaunchKernelGGL(loss_seg_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)pred, target, wmap, stats, sums, S, h, w, red_ws, red_counters_);
  }
  return check_launch("loss_reduce");
}

extern "C" int spg_loss_finalize(const float* stats, const float* seg_sums, const float* edge_sums, float* out, int B, int S,
                                 float sw0, float sw1, float sw2, float bce_w, float iou_w, float edge_w, spg_stream_t stream) {
  LossCfg c{{sw0, sw1, sw2}, bce_w, iou_w, edge_w, 0.f, 0.f};
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, seg_sums, edge_sums, out, B, (long)S * S, c);
  return check_launch("loss_finalize");
}

extern "C" int spg_loss_grad(int dtype, const void* pred, const float* target, const float* wmap, const float* stats,
                             const float* sums, const float* grad_out, void* dpred, int B, int S, int h, int w, int edge, float coef,
                             float bce_w, float iou_w, float alpha, float gamma, float* dz_ws, spg_stream_t stream) {
  SPG_REQUIRE(S % h == 0 && S % w == 0, "loss_grad: target size must be a multiple of the prediction size");
  if (dz_ws && !(h == S && w == S)) {   // two passes: dL/dz per full-res pixel, then the bilinear adjoint (B*S*S floats of scratch)
    hipStream_t s2 = (hipStream_t)stream;
    const long npx = (long)B * S * S;
    long g1 = (npx + 255) / 256;
    if (g1 > 8192) g1 = 8192;
    if (edge) {
      if (dtype == SPG_BF16) hipLaunchKernelGGL((loss_dz_kernel<bf16_t, true>), dim3((int)g1), dim3(256), 0, s2, (const bf16_t*)pred, target, wmap, stats, sums, dz_ws, B, S, h, w, bce_w, iou_w, alpha, gamma);
      else hipLaunchKernelGGL((loss_dz_kernel<float, true>), dim3((int)g1), dim3(256), 0, s2, (const float*)pred, target, wmap, stats, sums, dz_ws, B, S, h, w, bce_w, iou_w, alpha, gamma);
    } else {
      if (dtype == SPG_BF16) hipLaunchKernelGGL((loss_dz_kernel<bf16_t, false>), dim3((int)g1), dim3(256), 0, s2, (const bf16_t*)pred, target, wmap, stats, sums, dz_ws, B, S, h, w, bce_w, iou_w, alpha, gamma);
      else hipLaunchKernelGGL((loss_dz_kernel<float, false>), dim3((int)g1), dim3(256), 0, s2, (const float*)pred, target, wmap, stats, sums, dz_ws, B, S, h, w, bce_w, iou_w, alpha, gamma);
    }
    int rc = check_launch("loss_grad(dz)");
    if (rc) return rc;
    const int L = loss_gather_lanes(S, h, w);
    long g2 = ((long)B * h * w * L + 255) / 256;
    if (g2 > 8192) g2 = 8192;
#define SPG_GATHER(T_, L_) hipLaunchKernelGGL((loss_gather_kernel<T_, L_>), dim3((int)g2), dim3(256), 0, s2, dz_ws, grad_out, (T_*)dpred, B, S, h, w, coef)
    if (dtype == SPG_BF16) { if (L == 4) SPG_GATHER(bf16_t, 4); else if (L == 16) SPG_GATHER(bf16_t, 16); else SPG_GATHER(bf16_t, 64); }
    else { if (L == 4) SPG_GATHER(float, 4); else if (L == 16) SPG_GATHER(float, 16); else SPG_GATHER(float, 64); }
#undef SPG_GATHER
    return check_launch("loss_grad(gather)");
  }
  const long total = (long)B * h * w;   // one wave per low-res pixel (one thread when h == S)
  long g = (h == S && w == S) ? (total + 255) / 256 : (total + 3) / 4;
  if (g > 8192) g = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (edge) {
    if (dtype == SPG_BF16) hipLaunchKernelGGL((loss_grad_kernel<bf16_t, true>), dim3((int)g), dim3(256), 0, s, (const bf16_t*)pred, target, wmap, stats, sums, grad_out, (bf16_t*)dpred, B, S, h, w, coef, bce_w, iou_w, alpha, gamma);
    else hipLaunchKernelGGL((loss_grad_kernel<float, true>), dim3((int)g), dim3(256), 0, s, (const float*)pred, target, wmap, stats, sums, grad_out, (float*)dpred, B, S, h, w, coef, bce_w, iou_w, alpha, gamma);
  } else {
    if (dtype == SPG_BF16) hipLaunchKernelGGL((loss_grad_kernel<bf16_t, false>), dim3((int)g), dim3(256), 0, s, (const bf16_t*)pred, target, wmap, stats, sums, grad_out, (bf16_t*)dpred, B, S, h, w, coef, bce_w, iou_w, alpha, gamma);
    else hipLaunchKernelGGL((loss_grad_kernel<float, false>), dim3((int)g), dim3(256), 0, s, (const float*)pred, target, wmap, stats, sums, grad_out, (float*)dpred, B, S, h, w, coef, bce_w, iou_w, alpha, gamma);
  }
  return check_launch("loss_grad");
}

// ---- the four maps per launch ----------------------------------------------------------------------------------------------------------
static int loss_maps_fill(LossMaps& ms, const void* const* preds, void* const* dpreds, const int* hs, const int* ws, const float* coefs, int S,
                          const char* who) {
  for (int i = 0; i < 4; ++i) {
    SPG_REQUIRE(preds[i] && hs[i] > 0 && ws[i] > 0 && S % hs[i] == 0 && S % ws[i] == 0,
                "%s: map %d: target size %d must be a multiple of the prediction size %dx%d", who, i, S, hs[i], ws[i]);
    ms.pred[i] = preds[i]; ms.dpred[i] = dpreds ? dpreds[i] : nullptr;
    ms.h[i] = hs[i]; ms.w[i] = ws[i]; ms.coef[i] = coefs ? coefs[i] : 0.f;
    SPG_REQUIRE(!dpreds || dpreds[i], "%s: map %d has no gradient buffer", who, i);
  }
  return 0;
}
/* floats of reduction scratch / zeroed counters spg_loss_reduce_all needs */
extern "C" long spg_loss_reduce_all_workspace_floats(int B) { return 4L * B * LOSS_RED_BLOCKS * 4; }

extern "C" int spg_loss_reduce_all(int dtype, const void* const* preds, const int* hs, const int* ws, const float* masks, const float* edge_gt,
                                   const float* wmap, const float* stats, float* seg_sums, float* edge_sums, int B, int S, float alpha,
                                   float gamma, float* red_ws, long red_ws_floats, unsigned* red_counters_, spg_stream_t stream) {
  SPG_REQUIRE(B > 0 && S > 0 && masks && edge_gt && wmap && stats && seg_sums && edge_sums, "loss_reduce_all: null argument");
  SPG_REQUIRE(red_ws && red_counters_ && red_ws_floats >= spg_loss_reduce_all_workspace_floats(B), "loss_reduce_all: reduction workspace too small");
  LossMaps ms;
  if (int rc = loss_maps_fill(ms, preds, nullptr, hs, ws, nullptr, S, "loss_reduce_all")) return rc;
  long per = ((long)S * S + 255) / 256;
  dim3 grid((int)(per < LOSS_RED_BLOCKS ? per : LOSS_RED_BLOCKS), B, 4);     // (the same blocks per map as spg_loss_reduce: same partials, same sums)
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(loss_reduce_all_kernel<bf16_t>, grid, dim3(256), 0, s, ms, masks, edge_gt, wmap, stats, seg_sums, edge_sums, B, S, alpha, gamma, red_ws, red_counters_);
  else hipLaunchKernelGGL(loss_reduce_all_kernel<float>, grid, dim3(256), 0, s, ms, masks, edge_gt, wmap, stats, seg_sums, edge_sums, B, S, alpha, gamma, red_ws, red_counters_);
  return check_launch("loss_reduce_all");
}

/* dz_ws: 4 * B * S * S floats.  coefs[i]: scale weight / B for the three segmentation maps, edge weight / B for the edge map. */
extern "C" int spg_loss_grad_all(int dtype, const void* const* preds, void* const* dpreds, const int* hs, const int* ws, const float* coefs,
                                 const float* masks, const float* edge_gt, const float* wmap, const float* stats, const float* seg_sums,
                                 const float* edge_sums, const float* grad_out, int B, int S, float bce_w, float iou_w, float alpha, float gamma,
                                 float* dz_ws, spg_stream_t stream) {
  SPG_REQUIRE(B > 0 && S > 0 && masks && edge_gt && wmap && stats && seg_sums && edge_sums && dz_ws && coefs, "loss_grad_all: null argument");
  LossMaps ms;
  if (int rc = loss_maps_fill(ms, preds, dpreds, hs, ws, coefs, S, "loss_grad_all")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long npx = (long)B * S * S;
  long g1 = (npx + 255) / 256;
  if (g1 > 8192) g1 = 8192;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(loss_dz_all_kernel<bf16_t>, dim3((int)g1, 4), dim3(256), 0, s, ms, masks, edge_gt, wmap, stats, seg_sums, edge_sums, grad_out, dz_ws, B, S, bce_w, iou_w, alpha, gamma);
  else hipLaunchKernelGGL(loss_dz_all_kernel<float>, dim3((int)g1, 4), dim3(256), 0, s, ms, masks, edge_gt, wmap, stats, seg_sums, edge_sums, grad_out, dz_ws, B, S, bce_w, iou_w, alpha, gamma);
  if (int rc = check_launch("loss_grad_all(dz)")) return rc;
  long g2 = 0;
  for (int i = 0; i < 4; ++i) {
    if (hs[i] == S && ws[i] == S) continue;
    long g = ((long)B * hs[i] * ws[i] * loss_gather_lanes(S, hs[i], ws[i]) + 255) / 256;
    if (g > g2) g2 = g;
  }
  if (g2 == 0) return 0;                            // every map is at the target's resolution
  if (g2 > 8192) g2 = 8192;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(loss_gather_all_kernel<bf16_t>, dim3((int)g2, 4), dim3(256), 0, s, ms, dz_ws, grad_out, B, S);
  else hipLaunchKernelGGL(loss_gather_all_kernel<float>, dim3((int)g2, 4), dim3(256), 0, s, ms, dz_ws, grad_out, B, S);
  return check_launch("loss_grad_all(gather)");
}
