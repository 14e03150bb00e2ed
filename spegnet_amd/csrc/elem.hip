// HBM-bound element / gather kernels of the SPEGNet path (NHWC, 16-byte vectors over channels):
// bilinear upsample (+concat placement) and its adjoint, 2x2 max-pool with argmax, patch-embed im2col,
// SE excitation, dilated depth-wise 3x3, e-ASPP grouped fusion, 1x1 prediction heads, small utilities.
#include "common.h"

namespace spg {

static inline int ew_grid(long n_items) {
  long g = (n_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// PyTorch bilinear, align_corners=False: src = max(0,(dst+0.5)*in/out-0.5); i0=floor; i1=min(i0+1,in-1)
__device__ __forceinline__ void bil_src(int dst, int in, int out, int& i0, int& i1, float& lam) {
  float src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
  src = fmaxf(src, 0.f);
  i0 = min((int)src, in - 1);
  i1 = min(i0 + 1, in - 1);
  lam = src - (float)i0;
}

// y[b,Y,X,c0+c] = bilinear(x[b,:,:,c])  (y rows have ldy channels)
template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int h, int w,
                                                       int C, int H, int W, int ldy, int c0) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = (long)B * H * W * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    int y0, y1, x0, x1; float ly, lx;
    bil_src(Y, h, H, y0, y1, ly);
    bil_src(X, w, W, x0, x1, lx);
    const T* base = x + (long)b * h * w * C + ch * VEC;
    float a[VEC], bb[VEC], c[VEC], d[VEC], o[VEC];
    unpack16<T>(ld16(base + ((long)y0 * w + x0) * C), a);
    unpack16<T>(ld16(base + ((long)y0 * w + x1) * C), bb);
    unpack16<T>(ld16(base + ((long)y1 * w + x0) * C), c);
    unpack16<T>(ld16(base + ((long)y1 * w + x1) * C), d);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = w00 * a[e] + w01 * bb[e] + w10 * c[e] + w11 * d[e];
    st16(y + (((long)b * H + Y) * W + X) * ldy + c0 + ch * VEC, pack16<T>(o));
  }
}

// adjoint as a gather: dx[b,y,x,c] (+)= sum over output pixels whose taps hit (y,x)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int h,
                                                           int w, int C, int H, int W, int ldy, int c0, int accumulate) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = (long)B * h * w * nch;
  const int sy = (H + h - 1) / h, sx = (W + w - 1) / w;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int xx = (int)(p % w); p /= w;
    const int yy = (int)(p % h);
    const int b = (int)(p / h);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int Y0 = max(0, (yy - 1) * sy), Y1 = min(H - 1, (yy + 2) * sy);
    const int X0 = max(0, (xx - 1) * sx), X1 = min(W - 1, (xx + 2) * sx);
    for (int Y = Y0; Y <= Y1; ++Y) {
      int y0, y1; float ly;
      bil_src(Y, h, H, y0, y1, ly);
      const float wy = (y0 == yy ? 1.f - ly : 0.f) + (y1 == yy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int X = X0; X <= X1; ++X) {
        int x0, x1; float lx;
        bil_src(X, w, W, x0, x1, lx);
        const float wx = (x0 == xx ? 1.f - lx : 0.f) + (x1 == xx ? lx : 0.f);
        if (wx == 0.f) continue;
        float v[VEC];
        unpack16<T>(ld16(dy + (((long)b * H + Y) * W + X) * ldy + c0 + ch * VEC), v);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += wgt * v[e];
      }
    }
    T* dst = dx + i * VEC;
    if (accumulate) {
      float o[VEC];
      unpack16<T>(ld16(dst), o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += o[e];
    }
    st16(dst, pack16<T>(acc));
  }
}

// y[m][cy0 + c] (+)= x[m][cx0 + c]
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ x, T* __restrict__ y, long M, int C,
                                                            int ldx, int cx0, int ldy, int cy0, int accumulate) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = M * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    const long m = i / nch;
    u32x4 v = ld16(x + m * ldx + cx0 + ch * VEC);
    T* dst = y + m * ldy + cy0 + ch * VEC;
    if (accumulate) {
      float a[VEC], b[VEC];
      unpack16<T>(v, a);
      unpack16<T>(ld16(dst), b);
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] += b[e];
      v = pack16<T>(a);
    }
    st16(dst, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long nvec) {
  constexpr int VEC = ST<T>::VEC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float x[VEC], y[VEC];
    unpack16<T>(ld16(a + i * VEC), x);
    unpack16<T>(ld16(b + i * VEC), y);
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[e] += y[e];
    st16(o + i * VEC, pack16<T>(x));
  }
}

// f32 <-> T conversions (n multiple of 8)
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ f, bf16_t* __restrict__ h, long n8, int to_f32) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    if (to_f32) {
      float v[8];
      unpack16<bf16_t>(ld16(h + i * 8), v);
      float* dst = const_cast<float*>(f) + i * 8;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(f + i * 8), b = *reinterpret_cast<const f32x4*>(f + i * 8 + 4);
      float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      st16(h + i * 8, pack16<bf16_t>(v));
    }
  }
}

// bf16 -> f32 (n multiple of 8) AND the sum of squares of what was written, one partial per block (grid-stride walk, block_sum: a fixed
// order).  The gradient all-reduce's cast-back pass touches every element of the summed gradient anyway: the clip's norm needs no pass of
// its own (spg_sumsq_fold adds the partial arrays).
__global__ __launch_bounds__(256) void cast_sq_kernel(float* __restrict__ f, const bf16_t* __restrict__ h, long n8, float* __restrict__ sq_part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    unpack16<bf16_t>(ld16(h + i * 8), v);
    float* dst = f + i * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) sq_part[blockIdx.x] = s;
}

// 2x2 max pool over a channel window of an NHWC tensor; first max wins ties (torch semantics)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                           int B, int H, int W, int C, int ldc, int c0) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * Ho * Wo * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long p = i / C;
    const int X = (int)(p % Wo); p /= Wo;
    const int Y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const T* base = x + (((long)b * H + 2 * Y) * W + 2 * X) * ldc + c0 + c;
    float best = ST<T>::ld(base);
    int bi = 0;
    const float v1 = ST<T>::ld(base + ldc), v2 = ST<T>::ld(base + (long)W * ldc), v3 = ST<T>::ld(base + (long)(W + 1) * ldc);
    if (v1 > best) { best = v1; bi = 1; }
    if (v2 > best) { best = v2; bi = 2; }
    if (v3 > best) { best = v3; bi = 3; }
    ST<T>::st(y + i, best);
    idx[i] = (uint8_t)bi;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                           T* __restrict__ dx, int B, int H, int W, int C, int ldc, int c0) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long p = i / C;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    const long o = (((long)b * Ho + (Y >> 1)) * Wo + (X >> 1)) * C + c;
    const int me = ((Y & 1) << 1) | (X & 1);
    const float v = (idx[o] == me) ? ST<T>::ld(dy + o) : 0.f;
    ST<T>::st(dx + (((long)b * H + Y) * W + X) * ldc + c0 + c, v);
  }
}

// 16-byte versions (C, ldc, c0 multiples of the vector width): a thread owns VEC channels of one pooled pixel
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                               int B, int H, int W, int C, int ldc, int c0) {
  constexpr int VEC = ST<T>::VEC;
  const int Ho = H / 2, Wo = W / 2, nch = C / VEC;
  const long total = (long)B * Ho * Wo * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int X = (int)(p % Wo); p /= Wo;
    const int Y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const T* base = x + (((long)b * H + 2 * Y) * W + 2 * X) * ldc + c0 + ch * VEC;
    float v0[VEC], v1[VEC], v2[VEC], v3[VEC], best[VEC];
    unpack16<T>(ld16(base), v0); unpack16<T>(ld16(base + ldc), v1);
    unpack16<T>(ld16(base + (long)W * ldc), v2); unpack16<T>(ld16(base + (long)(W + 1) * ldc), v3);
    uint8_t bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      best[e] = v0[e]; bi[e] = 0;
      if (v1[e] > best[e]) { best[e] = v1[e]; bi[e] = 1; }
      if (v2[e] > best[e]) { best[e] = v2[e]; bi[e] = 2; }
      if (v3[e] > best[e]) { best[e] = v3[e]; bi[e] = 3; }
    }
    const long o = (((long)b * Ho + Y) * Wo + X) * C + ch * VEC;
    st16(y + o, pack16<T>(best));
#pragma unroll
    for (int e = 0; e < VEC; ++e) idx[o + e] = bi[e];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_vec_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int B, int H, int W, int C, int ldc, int c0) {
  constexpr int VEC = ST<T>::VEC;
  const int Ho = H / 2, Wo = W / 2, nch = C / VEC;
  const long total = (long)B * Ho * Wo * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int X = (int)(p % Wo); p /= Wo;
    const int Y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const long o = (((long)b * Ho + Y) * Wo + X) * C + ch * VEC;
    float g[VEC];
    unpack16<T>(ld16(dy + o), g);
    uint8_t bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bi[e] = idx[o + e];
    T* base = dx + (((long)b * H + 2 * Y) * W + 2 * X) * ldc + c0 + ch * VEC;
#pragma unroll
    for (int me = 0; me < 4; ++me) {
      float v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = bi[e] == me ? g[e] : 0.f;
      st16(base + ((long)(me >> 1) * W + (me & 1)) * ldc, pack16<T>(v));
    }
  }
}

// patch-embed im2col: img f32 NCHW [B,3,H,W] -> cols T [B*(H/4)*(W/4)][Kpad], k = c*49 + ky*7 + kx (conv 7x7 s4 p3).
// One workgroup = one output row segment of <= PI_PX pixels: the 3 x 7 input row segments it reads are staged in LDS with coalesced
// 16-byte loads (left halo of 4, zero rows outside the image), then every thread assembles 16-byte pieces of the column matrix from a
// k -> LDS-offset table, consecutive lanes = consecutive pieces (fully coalesced stores).  The element-per-thread version this replaces
// took 2.3 ms at batch 64 (41 G elements/s: 64-bit div/mod per element, 2-byte stores).
constexpr int PI_PX = 96;                  // output pixels per workgroup
constexpr int PI_RS = PI_PX * 4 + 4;       // LDS row: 4 halo floats (3 used) + 4 input pixels per output pixel
template <typename T>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int H, int W, int Kpad) {
  constexpr int VEC = ST<T>::VEC;
  __shared__ __attribute__((aligned(16))) float rows[21 * PI_RS + 4];
  __shared__ int tab[256];
  const int Ho = H / 4, Wo = W / 4;
  const int xt = (Wo + PI_PX - 1) / PI_PX;
  int blk = blockIdx.x;
  const int tx = blk % xt; blk /= xt;
  const int oy = blk % Ho;
  const int b = blk / Ho;
  const int ox0 = tx * PI_PX;
  const int npx = min(PI_PX, Wo - ox0);
  const int tid = threadIdx.x;
  // k -> offset of (c, ky, kx) in the staged rows (+1: column 4*ox+kx-3 sits at 4*ox+kx+1 behind the 4-float halo); pad columns read the zero slot
  if (tid < Kpad) {
    int off = 21 * PI_RS;
    if (tid < 147) { const int c = tid / 49, r = tid - c * 49, ky = r / 7, kx = r - ky * 7; off = (c * 7 + ky) * PI_RS + kx + 1; }
    tab[tid] = off;
  }
  if (tid < 4) rows[21 * PI_RS + tid] = 0.f;
  // stage: 21 rows x (1 halo quad + npx quads)
  const int quads = npx + 1;
  for (int i = tid; i < 21 * quads; i += 256) {
    const int r = i / quads, q = i - r * quads;
    const int c = r / 7, ky = r - c * 7;
    const int iy = oy * 4 + ky - 3;
    const int ix = ox0 * 4 + (q - 1) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)H && ix >= 0) v = *reinterpret_cast<const float4*>(img + (((long)b * 3 + c) * H + iy) * W + ix);
    *reinterpret_cast<float4*>(rows + r * PI_RS + q * 4) = v;
  }
  __syncthreads();
  const int cpp = Kpad / VEC;              // 16-byte pieces per pixel
  T* out = cols + (((long)b * Ho + oy) * Wo + ox0) * Kpad;
  for (int i = tid; i < npx * cpp; i += 256) {
    const int px = i / cpp, j = i - px * cpp;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int t = tab[j * VEC + e];
      v[e] = rows[t + (t < 21 * PI_RS ? px * 4 : 0)];
    }
    st16(out + (long)i * VEC, pack16<T>(v));
  }
}

// ---- SE excitation (per image): hidden = relu(W1 gap), scale = sigmoid(W2 hidden) ---------------------------
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ gap, const float* __restrict__ w1,
                                                    const float* __restrict__ w2, float* __restrict__ hidden,
                                                    float* __restrict__ scale, int C, int R, float in_scale) {
  extern __shared__ float sm[];  // gap[C] + hidden[R]
  const int b = blockIdx.x;
  float* g = sm; float* h = sm + C;
  for (int c = threadIdx.x; c < C; c += 256) g[c] = gap[(long)b * C + c] * in_scale;   // (in_scale = 1 / HW: gap holds column SUMS)
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (both matrices are cold fp32 master weights: the loads of up to 8 rows are issued together, one memory round trip per 32 rows
  // instead of one per row -- 30 -> ~12 us for C = 512, R = 32)
  for (int r0 = wave * 8; r0 < R; r0 += 32) {
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if ((C & 3) == 0) {
      for (int c = lane * 4; c < C; c += 256) {
        float4 wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = (r0 + j < R) ? *reinterpret_cast<const float4*>(w1 + (long)(r0 + j) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += wv[j].x * g[c] + wv[j].y * g[c + 1] + wv[j].z * g[c + 2] + wv[j].w * g[c + 3];
      }
    } else {
      for (int c = lane; c < C; c += 64)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (r0 + j < R) s[j] += w1[(long)(r0 + j) * C + c] * g[c];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum(s[j]);
      if (lane == 0 && r0 + j < R) { h[r0 + j] = fmaxf(t, 0.f); hidden[(long)b * R + r0 + j] = h[r0 + j]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    if ((R & 3) == 0) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int r = 0; r < R; r += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(w2 + (long)c * R + r);
        s0 += wv.x * h[r]; s1 += wv.y * h[r + 1]; s2 += wv.z * h[r + 2]; s3 += wv.w * h[r + 3];
      }
      s = (s0 + s1) + (s2 + s3);
    } else {
      for (int r = 0; r < R; ++r) s += w2[(long)c * R + r] * h[r];
    }
    scale[(long)b * C + c] = sigmoid_f(s);
  }
}
// given dscale -> dgap, dw1 +=, dw2 += (one block per image; the weight gradients sum over images in image order, by the block
// that arrives last: ws holds dz[B][C] then dh[B][R])
__global__ __launch_bounds__(1024) void se_fc_bwd_kernel(const float* __restrict__ gap, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ hidden,
                                                         const float* __restrict__ scale, const float* __restrict__ dscale,
                                                         float* __restrict__ dgap, float* __restrict__ dw1,
                                                         float* __restrict__ dw2, int C, int R, float* __restrict__ ws,
                                                         unsigned* __restrict__ counter, float in_scale, int bchunk) {
  extern __shared__ float sm[];  // dz[C] + dh[R]; the last block re-uses it as gdz[B][C] | gap[B][C] | gdh[B][R] | hidden[B][R]
  __shared__ unsigned s_last;
  const int b = blockIdx.x, B = gridDim.x, nt = (int)blockDim.x;
  float* dz = sm; float* dh = sm + C;
  float* gdz = ws; float* gdh = ws + (long)B * C;
  for (int c = threadIdx.x; c < C; c += nt) {
    const float s = scale[(long)b * C + c];
    dz[c] = dscale[(long)b * C + c] * s * (1.f - s);
    st_part(gdz + (long)b * C + c, dz[c]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w2[(long)c * R + r] * dz[c];
    s = wave_sum(s);
    const float hr = hidden[(long)b * R + r];
    if (lane == 0) { dh[r] = hr > 0.f ? s : 0.f; st_part(gdh + (long)b * R + r, dh[r]); }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += nt) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += w1[(long)r * C + c] * dh[r];
    dgap[(long)b * C + c] = s;
  }
  if (!arrive_last(counter, (unsigned)B, &s_last)) return;
  // the last block: weight gradients summed over the images in image order, operands staged in LDS (coalesced) first -- `bchunk` images
  // at a time (what 64 KiB of LDS hold: one chunk up to batch 15 at C = 512, R = 32; larger batches add chunk after chunk, still in image order)
  for (int b0 = 0; b0 < B; b0 += bchunk) {
    const int nb = min(bchunk, B - b0);
    float* l_dz = sm; float* l_gap = l_dz + nb * C; float* l_dh = l_gap + nb * C; float* l_hid = l_dh + nb * R;
    __syncthreads();                                        // (the previous chunk's readers are done with the staging area)
    for (int i = threadIdx.x; i < nb * C; i += nt) { l_dz[i] = gdz[(long)b0 * C + i]; l_gap[i] = gap[(long)b0 * C + i] * in_scale; }
    for (int i = threadIdx.x; i < nb * R; i += nt) { l_dh[i] = gdh[(long)b0 * R + i]; l_hid[i] = hidden[(long)b0 * R + i]; }
    __syncthreads();
    // (the old gradient values of eight elements per thread are loaded together: `dw[i] += s` in a loop is one memory round trip per element)
    for (int i0 = threadIdx.x; i0 < C * R; i0 += nt * 8) {   // dw2 [C][R]: consecutive threads, consecutive r
      float o2[8], o1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = i0 + j * nt; o2[j] = i < C * R ? dw2[i] : 0.f; o1[j] = i < C * R ? dw1[i] : 0.f; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * nt;
        if (i < C * R) {
          const int c = i / R, r = i - c * R;
          float s2 = 0.f;
          for (int bb = 0; bb < nb; ++bb) s2 += l_dz[bb * C + c] * l_hid[bb * R + r];
          dw2[i] = o2[j] + s2;
          const int r1 = i / C, c1 = i - r1 * C;             // dw1 [R][C]: consecutive threads, consecutive c
          float s1 = 0.f;
          for (int bb = 0; bb < nb; ++bb) s1 += l_dh[bb * R + r1] * l_gap[bb * C + c1];
          dw1[i] = o1[j] + s1;
        }
      }
    }
  }
}

// y = x * scale[b][c]
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                         T* __restrict__ y, long HW, int C, long total) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    const long b = (i / nch) / HW;
    float v[VEC];
    unpack16<T>(ld16(x + i * VEC), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] *= scale[b * C + ch * VEC + e];
    st16(y + i * VEC, pack16<T>(v));
  }
}
// dx = dy*scale[b][c] + dgap_over_hw[b][c]
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ scale,
                                                             const float* __restrict__ dgap, T* __restrict__ dx, long HW,
                                                             int C, long total, float inv_hw) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    const long b = (i / nch) / HW;
    float v[VEC];
    unpack16<T>(ld16(dy + i * VEC), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = v[e] * scale[b * C + ch * VEC + e] + dgap[b * C + ch * VEC + e] * inv_hw;
    st16(dx + i * VEC, pack16<T>(v));
  }
}

// ---- dilated depth-wise 3x3 (pad = dil), NHWC, weights f32 [C][9] ------------------------------------------
// A thread keeps ONE 16-byte channel chunk for its whole grid-stride walk (256 and the grid stride are multiples of the chunk count),
// so its 9 x VEC weights are loaded once into registers instead of 72 scalar loads per pixel.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                     int B, int H, int W, int C, int dil, int flip) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = (long)B * H * W * nch;
  const bool fixed = (256 % nch) == 0;                 // chunk index is loop invariant
  float wr[9][VEC];
  const int ch0 = threadIdx.x % nch;
  if (fixed) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < VEC; ++e) wr[t][e] = w[(ch0 * VEC + e) * 9 + (flip ? 8 - t : t)];
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    long p = i / nch;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = (t / 3 - 1) * dil, dx = (t % 3 - 1) * dil;
      const int yy = Y + dy, xx = X + dx;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        float v[VEC];
        unpack16<T>(ld16(x + (((long)b * H + yy) * W + xx) * C + ch * VEC), v);
        if (fixed) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += wr[t][e] * v[e];
        } else {
          const int tw = flip ? 8 - t : t;
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += w[(ch * VEC + e) * 9 + tw] * v[e];
        }
      }
    }
    st16(y + i * VEC, pack16<T>(acc));
  }
}
// dw[c][t] += sum_pixels dy[p][c] * x[p + t][c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           float* __restrict__ dw, int B, int H, int W, int C, int dil,
                                                           long pix_per_block, float* __restrict__ part, unsigned* __restrict__ counter) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int ppar = 256 / nch;
  const int ch = threadIdx.x % nch, pl = threadIdx.x / nch;
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
      float d[VEC];
      unpack16<T>(ld16(dy + p * C + ch * VEC), d);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = Y + (t / 3 - 1) * dil, xx = X + (t % 3 - 1) * dil;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          float v[VEC];
          unpack16<T>(ld16(x + (p + (long)(t / 3 - 1) * dil * W + (t % 3 - 1) * dil) * C + ch * VEC), v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[t][e] += d[e] * v[e];
        }
      }
    }
  }
  __shared__ float red[256 * 8];
  __shared__ unsigned s_last;
  float* mypart = part + (long)blockIdx.x * 9 * C;
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
  if (!arrive_last(counter, gridDim.x, &s_last)) return;
  finish_partials<256>(part, gridDim.x, 9 * C, 9 * C, dw, 1, red);
}

// ---- e-ASPP grouped 1x1 over the branch-major concat: y[p][g] = sum_j w[g][j] * cat[p][5g+j] --------------
template <typename T>
__device__ __forceinline__ float cat_at(const T* const* br, const float* glob, long b, long p, int C, int cc) {
  const int which = cc / C, ch = cc - which * C;
  return which < 4 ? ST<T>::ld(br[which] + p * C + ch) : glob[b * C + ch];
}
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_kernel(const T* b0, const T* b1, const T* b2, const T* b3,
                                                         const float* __restrict__ glob, const float* __restrict__ w,
                                                         T* __restrict__ y, long HW, int C, long total) {
  const T* br[4] = {b0, b1, b2, b3};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i % C);
    const long p = i / C;
    const long b = p / HW;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) s += w[g * 5 + j] * cat_at<T>(br, glob, b, p, C, 5 * g + j);
    ST<T>::st(y + i, s);
  }
}
// d(cat)[p][cc] = w[cc/5][cc%5] * dy[p][cc/5]  -> written to the four branch grads; global part reduced per image
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                             T* d0, T* d1, T* d2, T* d3, long HW, int C, long total) {
  T* d[4] = {d0, d1, d2, d3};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {  // over p * 4C
    const int cc = (int)(i % (4 * C));
    const long p = i / (4 * C);
    const int g = cc / 5, j = cc - g * 5;
    const int which = cc / C, ch = cc - which * C;
    ST<T>::st(d[which] + p * C + ch, w[g * 5 + j] * ST<T>::ld(dy + p * C + g));
  }
}
// per image: dglob[b][ch] = sum_p w[g][j]*dy[p][g] for cc = 4C+ch;   dw[g][j] += sum_p dy[p][g]*cat[p][5g+j]
template <typename T>
__global__ __launch_bounds__(256) void easpp_fuse_bwd_reduce_kernel(const T* __restrict__ dy, const T* b0, const T* b1,
                                                                    const T* b2, const T* b3, const float* __restrict__ glob,
                                                                    const float* __restrict__ w, float* __restrict__ dglob,
                                                                    float* __restrict__ dw, long HW, int C, long rows_per_block,
                                                                    float* __restrict__ part, unsigned* __restrict__ counter) {
  const T* br[4] = {b0, b1, b2, b3};
  __shared__ float scratch[256];
  __shared__ unsigned s_last;
  float* mypart = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 6 * C;   // [5C: dw partials][C: dglob partials]
  const long b = blockIdx.y;
  const long r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  // thread -> (g, j) pairs: 5*C of them; loop
  for (int gj = threadIdx.x; gj < 5 * C; gj += 256) {
    const int g = gj / 5, j = gj - g * 5, cc = gj;  // cc == 5g + j
    float sw = 0.f, sd = 0.f;
    for (long r = r0; r < r1; ++r) {
      const long p = b * HW + r;
      const float d = ST<T>::ld(dy + p * C + g);
      sw += d * cat_at<T>(br, glob, b, p, C, cc);
      sd += d;
    }
    st_part(mypart + gj, sw);
    if (cc >= 4 * C) st_part(mypart + 5 * C + (cc - 4 * C), w[gj] * sd);
  }
  const int gx = gridDim.x, nimg = gridDim.y;
  if (!arrive_last(counter, (unsigned)(gx * nimg), &s_last)) return;
  finish_partials<256>(part, gx * nimg, 6 * C, 5 * C, dw, 1, scratch);
  for (int bb = 0; bb < nimg; ++bb)
    finish_partials<256>(part + (long)bb * gx * 6 * C + 5 * C, gx, 6 * C, C, dglob + (long)bb * C, 0, scratch);
}

// ---- 1x1 prediction heads: y[m] = x[m,:].w + b --------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, long M, int C) {
  constexpr int VEC = ST<T>::VEC;
  const int lpp = C / VEC;  // lanes per pixel (power of two <= 64)
  const int ppw = 64 / lpp;
  const int lane = threadIdx.x & 63;
  const int sub = lane % lpp, pin = lane / lpp;
  const long wave_global = (blockIdx.x * 256L + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;
  float wv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) wv[e] = w[sub * VEC + e];
  for (long m0 = wave_global * ppw; m0 < M; m0 += nwaves * ppw) {
    const long m = m0 + pin;
    float s = 0.f;
    if (m < M) {
      float v[VEC];
      unpack16<T>(ld16(x + m * C + sub * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += v[e] * wv[e];
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && m < M) ST<T>::st(y + m, s + bias[0]);
  }
}
// dx[m,c] (+)= dy[m]*w[c]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                          T* __restrict__ dx, long M, int C, int accumulate) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const long total = M * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    const long m = i / nch;
    const float d = ST<T>::ld(dy + m);
    float v[VEC];
    if (accumulate) unpack16<T>(ld16(dx + i * VEC), v);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] += d * w[ch * VEC + e];
    st16(dx + i * VEC, pack16<T>(v));
  }
}
// dw[c] += sum_m dy[m]*x[m,c];  db += sum_m dy[m]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          float* __restrict__ dw, float* __restrict__ db, long M, int C,
                                                          long rows_per_block, float* __restrict__ part, unsigned* __restrict__ counter) {
  constexpr int VEC = ST<T>::VEC;
  const int nch = C / VEC;
  const int rpar = 256 / nch;
  const int ch = threadIdx.x % nch, rl = threadIdx.x / nch;
  const long r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc[VEC], sb = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (rl < rpar) {
    for (long r = r0 + rl; r < r1; r += rpar) {
      const float d = ST<T>::ld(dy + r);
      float v[VEC];
      unpack16<T>(ld16(x + r * C + ch * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += d * v[e];
      if (ch == 0) sb += d;
    }
  }
  __shared__ float red[256 * 8];
  __shared__ float redb[256];
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = acc[e];
  redb[threadIdx.x] = sb;
  __syncthreads();
  __shared__ unsigned s_last;
  float* mypart = part + (long)blockIdx.x * (C + 1);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = 0; r < rpar; ++r) s += red[r * nch * VEC + c];
    st_part(mypart + c, s);
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < 256; ++r) s += redb[r];
    st_part(mypart + C, s);
  }
  if (!arrive_last(counter, gridDim.x, &s_last)) return;
  finish_partials<256>(part, gridDim.x, C + 1, C, dw, 1, red);
  finish_partials<256>(part + C, gridDim.x, C + 1, 1, db, 1, red);
}


// dst[r][0..Kp) = cast(a[r][0..na) | b[r][0..nb) | zeros): the position-embedding GEMM's weight operand [pos_embed | pos_embed_window]
template <typename T>
__global__ __launch_bounds__(256) void pack_cols2_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                         T* __restrict__ dst, int R, int Kp) {
  const long total = (long)R * Kp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / Kp), k = (int)(i - (long)r * Kp);
    const float v = k < na ? a[(long)r * na + k] : (k < na + nb ? b[(long)r * nb + (k - na)] : 0.f);
    ST<T>::st(dst + i, v);
  }
}
// up to 4 jobs of dst[r][c] += src[r][c] (c < C; row strides ldd / lds) in one launch: column slices of padded wgrad results into
// their parameters' gradients
struct AddColsJob { float* dst; const float* src; int R, C, ldd, lds; };
struct AddColsBatch { AddColsJob job[4]; int njobs; };
__global__ __launch_bounds__(256) void add_cols_batch_kernel(AddColsBatch bt) {
  const AddColsJob& j = bt.job[blockIdx.y];
  const long total = (long)j.R * j.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / j.C), c = (int)(i - (long)r * j.C);
    j.dst[(long)r * j.ldd + c] += j.src[(long)r * j.lds + c];
  }
}
}  // namespace spg

using namespace spg;

#define LAUNCH_T(dtype, KERNEL, grid, lds, stream, ...)                                                      \
  do {                                                                                                       \
    if ((dtype) == SPG_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, dim3(grid), dim3(256), lds, (hipStream_t)(stream), __VA_ARGS__);  \
  } while (0)
#define TP(dtype, p) ((dtype) == SPG_BF16 ? (void*)(p) : (void*)(p))
static inline int vec_of(int dtype) { return dtype == SPG_BF16 ? 8 : 4; }

extern "C" int spg_upsample_bilinear(int dtype, const void* x, void* y, int B, int h, int w, int C, int H, int W, int ldy,
                                     int c0, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0 && ldy % v == 0 && c0 % v == 0, "upsample: C=%d ldy=%d c0=%d must be multiples of %d", C, ldy, c0, v);
  const int grid = ew_grid((long)B * H * W * (C / v));
  if (dtype == SPG_BF16) hipLaunchKernelGGL(upsample_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, h, w, C, H, W, ldy, c0);
  else hipLaunchKernelGGL(upsample_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, h, w, C, H, W, ldy, c0);
  return check_launch("upsample_bilinear");
}
extern "C" int spg_upsample_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int h, int w, int C, int H, int W,
                                         int ldy, int c0, int accumulate, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0 && ldy % v == 0 && c0 % v == 0, "upsample_bwd: alignment");
  SPG_REQUIRE(H >= h && W >= w, "upsample_bwd: only upsampling is supported");
  const int grid = ew_grid((long)B * h * w * (C / v));
  if (dtype == SPG_BF16) hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
  else hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, B, h, w, C, H, W, ldy, c0, accumulate);
  return check_launch("upsample_bilinear_bwd");
}
extern "C" int spg_copy_channels(int dtype, const void* x, void* y, long M, int C, int ldx, int cx0, int ldy, int cy0,
                                 int accumulate, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0 && ldx % v == 0 && ldy % v == 0 && cx0 % v == 0 && cy0 % v == 0, "copy_channels: alignment");
  const int grid = ew_grid(M * (C / v));
  if (dtype == SPG_BF16) hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, M, C, ldx, cx0, ldy, cy0, accumulate);
  else hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, M, C, ldx, cx0, ldy, cy0, accumulate);
  return check_launch("copy_channels");
}
extern "C" int spg_add(int dtype, const void* a, const void* b, void* out, long n, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(n % v == 0, "add: n=%ld must be a multiple of %d", n, v);
  const int grid = ew_grid(n / v);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / v);
  else hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)out, n / v);
  return check_launch("add");
}
extern "C" int spg_cast_bf16(const float* f32, void* bf16, long n, int to_f32, spg_stream_t stream) {
  SPG_REQUIRE(n % 8 == 0, "cast: n=%ld must be a multiple of 8", n);
  hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, f32, (bf16_t*)bf16, n / 8, to_f32);
  return check_launch("cast_bf16");
}
extern "C" int spg_cast_bf16_sq(float* f32, const void* bf16, long n, float* sq_part, int nparts, spg_stream_t stream) {
  SPG_REQUIRE(n % 8 == 0 && n > 0, "cast_bf16_sq: n=%ld must be a positive multiple of 8", n);
  SPG_REQUIRE(sq_part && nparts >= 1 && nparts <= 4096, "cast_bf16_sq: 1..4096 partials");
  hipLaunchKernelGGL(cast_sq_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, f32, (const bf16_t*)bf16, n / 8, sq_part);
  return check_launch("cast_bf16_sq");
}
extern "C" int spg_maxpool2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int ldc, int c0,
                                spg_stream_t stream) {
  SPG_REQUIRE((H % 2 == 0) && (W % 2 == 0), "maxpool2: H=%d W=%d must be even", H, W);
  const int vw = vec_of(dtype);
  if (C % vw == 0 && ldc % vw == 0 && c0 % vw == 0) {
    const int gridv = ew_grid((long)B * (H / 2) * (W / 2) * (C / vw));
    if (dtype == SPG_BF16) hipLaunchKernelGGL(maxpool2_fwd_vec_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, idx, B, H, W, C, ldc, c0);
    else hipLaunchKernelGGL(maxpool2_fwd_vec_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, idx, B, H, W, C, ldc, c0);
    return check_launch("maxpool2_fwd(vec)");
  }
  const int grid = ew_grid((long)B * (H / 2) * (W / 2) * C);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, idx, B, H, W, C, ldc, c0);
  else hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, idx, B, H, W, C, ldc, c0);
  return check_launch("maxpool2_fwd");
}
extern "C" int spg_maxpool2_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, int ldc,
                                int c0, spg_stream_t stream) {
  SPG_REQUIRE((H % 2 == 0) && (W % 2 == 0), "maxpool2_bwd: H=%d W=%d must be even", H, W);
  const int vw = vec_of(dtype);
  if (C % vw == 0 && ldc % vw == 0 && c0 % vw == 0) {
    const int gridv = ew_grid((long)B * (H / 2) * (W / 2) * (C / vw));
    if (dtype == SPG_BF16) hipLaunchKernelGGL(maxpool2_bwd_vec_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, idx, (bf16_t*)dx, B, H, W, C, ldc, c0);
    else hipLaunchKernelGGL(maxpool2_bwd_vec_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)dy, idx, (float*)dx, B, H, W, C, ldc, c0);
    return check_launch("maxpool2_bwd(vec)");
  }
  const int grid = ew_grid((long)B * H * W * C);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, idx, (bf16_t*)dx, B, H, W, C, ldc, c0);
  else hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, idx, (float*)dx, B, H, W, C, ldc, c0);
  return check_launch("maxpool2_bwd");
}
extern "C" int spg_patch_im2col(int dtype, const float* img, void* cols, int B, int H, int W, int Kpad, spg_stream_t stream) {
  SPG_REQUIRE(B > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && Kpad >= 147 && Kpad <= 256 && Kpad % 8 == 0, "patch_im2col: B=%d H=%d W=%d Kpad=%d", B, H, W, Kpad);
  SPG_REQUIRE(((uintptr_t)img & 15) == 0 && ((uintptr_t)cols & 15) == 0, "patch_im2col: img / cols must be 16-byte aligned");
  const long nblk = (long)B * (H / 4) * ((W / 4 + PI_PX - 1) / PI_PX);
  SPG_REQUIRE(nblk < (1L << 31), "patch_im2col: %ld workgroups", nblk);
  const int grid = (int)nblk;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(patch_im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)cols, B, H, W, Kpad);
  else hipLaunchKernelGGL(patch_im2col_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (float*)cols, B, H, W, Kpad);
  return check_launch("patch_im2col");
}
// ---------------------------------------------------------------------------------------------------
// Input pipeline on the device: uint8 HWC image -> float / 255 -> antialiased bilinear resize -> (v - mean) / std
// in one kernel; output f32 [3, OH, OW] (the reference's layout, utils/image_processor.py:118-131).  The resize is ATen's
// _upsample_bilinear2d_aa (what F.interpolate(mode='bilinear', align_corners=False, antialias=True) runs): separable triangle
// filter whose support grows with the down-scale factor; horizontal pass first, then vertical, weights normalised per output index.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void aa_taps(int i, float scale, int in_size, int& lo, int& n, float& center, float& invscale) {
  const float support = scale >= 1.f ? scale : 1.f;
  invscale = scale >= 1.f ? 1.f / scale : 1.f;
  center = scale * ((float)i + 0.5f);
  lo = max((int)(center - support + 0.5f), 0);
  n = min((int)(center + support + 0.5f), in_size) - lo;
}
__device__ __forceinline__ float aa_tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

__global__ __launch_bounds__(256) void preprocess_image_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W, int OH,
                                                               int OW, float m0, float m1, float m2, float is0, float is1, float is2) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= OH * OW) return;
  const int oy = idx / OW, ox = idx - oy * OW;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  int ylo, yn, xlo, xn;
  float yc, yinv, xc, xinv;
  aa_taps(oy, sy, H, ylo, yn, yc, yinv);
  aa_taps(ox, sx, W, xlo, xn, xc, xinv);
  float wxs = 0.f, wys = 0.f;
  for (int i = 0; i < xn; ++i) wxs += aa_tri(((float)(i + xlo) - xc + 0.5f) * xinv);
  for (int j = 0; j < yn; ++j) wys += aa_tri(((float)(j + ylo) - yc + 0.5f) * yinv);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = 0; j < yn; ++j) {
    const float wy = aa_tri(((float)(j + ylo) - yc + 0.5f) * yinv) / wys;
    const uint8_t* row = img + ((long)(ylo + j) * W + xlo) * 3;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;                      // the horizontal pass of this input row
    for (int i = 0; i < xn; ++i) {
      const float wx = aa_tri(((float)(i + xlo) - xc + 0.5f) * xinv) / wxs;
      r0 += wx * ((float)row[3 * i] / 255.f);
      r1 += wx * ((float)row[3 * i + 1] / 255.f);
      r2 += wx * ((float)row[3 * i + 2] / 255.f);
    }
    a0 += wy * r0; a1 += wy * r1; a2 += wy * r2;
  }
  const long plane = (long)OH * OW;
  out[idx] = (a0 - m0) * is0;
  out[plane + idx] = (a1 - m1) * is1;
  out[2 * plane + idx] = (a2 - m2) * is2;
}

// The same arithmetic for a whole batch in ONE launch: images of different sizes packed in one device buffer (image i starts at byte
// offs[i]), blockIdx.y = image; output f32 [B, 3, OH, OW] -- the model's input batch.
constexpr int PRE_BATCH_MAX = 64;
struct PreBatch { long off[PRE_BATCH_MAX]; int H[PRE_BATCH_MAX], W[PRE_BATCH_MAX]; };
__global__ __launch_bounds__(256) void preprocess_batch_kernel(const uint8_t* __restrict__ base, PreBatch pb, float* __restrict__ out, int OH, int OW,
                                                               float m0, float m1, float m2, float is0, float is1, float is2) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= OH * OW) return;
  const int b = blockIdx.y;
  const int H = pb.H[b], W = pb.W[b];
  const uint8_t* img = base + pb.off[b];
  const int oy = idx / OW, ox = idx - oy * OW;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  int ylo, yn, xlo, xn;
  float yc, yinv, xc, xinv;
  aa_taps(oy, sy, H, ylo, yn, yc, yinv);
  aa_taps(ox, sx, W, xlo, xn, xc, xinv);
  float wxs = 0.f, wys = 0.f;
  for (int i = 0; i < xn; ++i) wxs += aa_tri(((float)(i + xlo) - xc + 0.5f) * xinv);
  for (int j = 0; j < yn; ++j) wys += aa_tri(((float)(j + ylo) - yc + 0.5f) * yinv);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = 0; j < yn; ++j) {
    const float wy = aa_tri(((float)(j + ylo) - yc + 0.5f) * yinv) / wys;
    const uint8_t* row = img + ((long)(ylo + j) * W + xlo) * 3;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int i = 0; i < xn; ++i) {
      const float wx = aa_tri(((float)(i + xlo) - xc + 0.5f) * xinv) / wxs;
      r0 += wx * ((float)row[3 * i] / 255.f);
      r1 += wx * ((float)row[3 * i + 1] / 255.f);
      r2 += wx * ((float)row[3 * i + 2] / 255.f);
    }
    a0 += wy * r0; a1 += wy * r1; a2 += wy * r2;
  }
  const long plane = (long)OH * OW;
  float* o = out + (long)b * 3 * plane;
  o[idx] = (a0 - m0) * is0;
  o[plane + idx] = (a1 - m1) * is1;
  o[2 * plane + idx] = (a2 - m2) * is2;
}

/* offs / H / W: HOST arrays of B entries (B <= 64 per call); base: device buffer holding the uint8 HWC images */
extern "C" int spg_preprocess_batch(const uint8_t* base, const long* offs, const int* H, const int* W, float* out_b3hw, int B, int OH, int OW,
                                    const float* mean3, const float* std3, spg_stream_t stream) {
  SPG_REQUIRE(B > 0 && B <= PRE_BATCH_MAX && OH > 0 && OW > 0 && base && offs && H && W && out_b3hw && mean3 && std3,
              "preprocess_batch: 1..%d images per call, got %d", PRE_BATCH_MAX, B);
  SPG_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "preprocess_batch: zero std");
  PreBatch pb;
  for (int i = 0; i < B; ++i) {
    SPG_REQUIRE(H[i] > 0 && W[i] > 0 && offs[i] >= 0, "preprocess_batch: image %d has size %dx%d", i, H[i], W[i]);
    pb.off[i] = offs[i]; pb.H[i] = H[i]; pb.W[i] = W[i];
  }
  for (int i = B; i < PRE_BATCH_MAX; ++i) { pb.off[i] = 0; pb.H[i] = 1; pb.W[i] = 1; }
  hipLaunchKernelGGL(preprocess_batch_kernel, dim3(cdiv((long)OH * OW, 256), B), dim3(256), 0, (hipStream_t)stream, base, pb, out_b3hw, OH, OW,
                     mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  return check_launch("preprocess_batch");
}

extern "C" int spg_preprocess_image(const uint8_t* img_hwc, float* out_chw, int H, int W, int OH, int OW, const float* mean3,
                                    const float* std3, spg_stream_t stream) {
  SPG_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && mean3 && std3, "preprocess_image: bad sizes %dx%d -> %dx%d", H, W, OH, OW);
  SPG_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "preprocess_image: zero std");
  hipLaunchKernelGGL(preprocess_image_kernel, dim3(cdiv((long)OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, img_hwc, out_chw, H, W, OH, OW,
                     mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  return check_launch("preprocess_image");
}

extern "C" int spg_pack_cols2(int dtype, const float* a, int na, const float* b, int nb, void* dst, int R, int Kp, spg_stream_t stream) {
  SPG_REQUIRE(R > 0 && na >= 0 && nb >= 0 && na + nb <= Kp, "pack_cols2: %d + %d columns do not fit %d", na, nb, Kp);
  const long total = (long)R * Kp;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(pack_cols2_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, (bf16_t*)dst, R, Kp);
  else hipLaunchKernelGGL(pack_cols2_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, (float*)dst, R, Kp);
  return check_launch("pack_cols2");
}
extern "C" int spg_add_cols_batch(int njobs, float* const* dst, const float* const* src, const int* R, const int* C, const int* ldd,
                                  const int* lds, spg_stream_t stream) {
  SPG_REQUIRE(njobs >= 1 && njobs <= 4, "add_cols_batch: 1..4 jobs, got %d", njobs);
  AddColsBatch bt;
  long mx = 0;
  for (int i = 0; i < njobs; ++i) {
    SPG_REQUIRE(R[i] > 0 && C[i] > 0 && ldd[i] >= C[i] && lds[i] >= C[i], "add_cols_batch: job %d: bad extents", i);
    bt.job[i] = AddColsJob{dst[i], src[i], R[i], C[i], ldd[i], lds[i]};
    mx = (long)R[i] * C[i] > mx ? (long)R[i] * C[i] : mx;
  }
  for (int i = njobs; i < 4; ++i) bt.job[i] = bt.job[0];
  bt.njobs = njobs;
  hipLaunchKernelGGL(add_cols_batch_kernel, dim3(ew_grid(mx), njobs), dim3(256), 0, (hipStream_t)stream, bt);
  return check_launch("add_cols_batch");
}
extern "C" int spg_se_fc(const float* gap, const float* w1, const float* w2, float* hidden, float* scale, int B, int C, int R,
                         float in_scale, spg_stream_t stream) {
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), (hipStream_t)stream, gap, w1, w2, hidden, scale, C, R, in_scale);
  return check_launch("se_fc");
}
extern "C" int spg_se_fc_bwd(const float* gap, const float* w1, const float* w2, const float* hidden, const float* scale,
                             const float* dscale, float* dgap, float* dw1, float* dw2, int B, int C, int R, float in_scale, float* red_ws,
                             long red_ws_floats, unsigned* red_counter, spg_stream_t stream) {
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= (long)B * (C + R), "se_fc_bwd: needs B*(C+R) floats of scratch and one zeroed counter");
  int bchunk = (int)((64 * 1024) / ((size_t)2 * (C + R) * sizeof(float)));     // images whose dz / gap / dh / hidden rows fit 64 KiB of LDS
  SPG_REQUIRE(bchunk >= 1, "se_fc_bwd: C + R = %d: one image's rows exceed the 64 KiB the last block stages", C + R);
  if (bchunk > B) bchunk = B;
  const size_t lds = (size_t)2 * bchunk * (C + R) * sizeof(float);     // (>= the C + R floats of the per-image phase)
  hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, gap, w1, w2, hidden, scale, dscale, dgap, dw1, dw2, C, R,
                     red_ws, red_counter, in_scale, bchunk);
  return check_launch("se_fc_bwd");
}
extern "C" int spg_chan_scale(int dtype, const void* x, const float* scale, void* y, int B, long HW, int C, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0, "chan_scale: C alignment");
  const long total = (long)B * HW * (C / v);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(chan_scale_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, (bf16_t*)y, HW, C, total);
  else hipLaunchKernelGGL(chan_scale_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, (float*)y, HW, C, total);
  return check_launch("chan_scale");
}
extern "C" int spg_chan_scale_bwd(int dtype, const void* dy, const float* scale, const float* dgap, void* dx, int B, long HW,
                                  int C, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0, "chan_scale_bwd: C alignment");
  const long total = (long)B * HW * (C / v);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(chan_scale_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, scale, dgap, (bf16_t*)dx, HW, C, total, 1.f / (float)HW);
  else hipLaunchKernelGGL(chan_scale_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, scale, dgap, (float*)dx, HW, C, total, 1.f / (float)HW);
  return check_launch("chan_scale_bwd");
}
extern "C" int spg_dwconv3x3(int dtype, const void* x, const float* w, void* y, int B, int H, int W, int C, int dil, int flip,
                             spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0, "dwconv3x3: C alignment");
  const int grid = ew_grid((long)B * H * W * (C / v));
  if (dtype == SPG_BF16) hipLaunchKernelGGL(dwconv_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)y, B, H, W, C, dil, flip);
  else hipLaunchKernelGGL(dwconv_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, (float*)y, B, H, W, C, dil, flip);
  return check_launch("dwconv3x3");
}
constexpr int DWW_MAX_BLOCKS = 64;
extern "C" int spg_dwconv3x3_wgrad(int dtype, const void* dy, const void* x, float* dw, int B, int H, int W, int C, int dil,
                                   float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0 && C / v <= 256, "dwconv3x3_wgrad: C alignment");
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= (long)DWW_MAX_BLOCKS * 9 * C, "dwconv3x3_wgrad: needs 64*9*C floats of scratch and one zeroed counter");
  const long npix = (long)B * H * W;
  long ppb = cdiv(npix, DWW_MAX_BLOCKS);
  if (ppb < 64) ppb = 64;
  const int grid = cdiv(npix, ppb);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, dw, B, H, W, C, dil, ppb, red_ws, red_counter);
  else hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, dw, B, H, W, C, dil, ppb, red_ws, red_counter);
  return check_launch("dwconv3x3_wgrad");
}
extern "C" int spg_easpp_fuse(int dtype, const void* br0, const void* br1, const void* br2, const void* br3, const float* glob,
                              const float* w, void* y, int B, long HW, int C, spg_stream_t stream) {
  const long total = (long)B * HW * C;
  if (dtype == SPG_BF16) hipLaunchKernelGGL(easpp_fuse_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)br0, (const bf16_t*)br1, (const bf16_t*)br2, (const bf16_t*)br3, glob, w, (bf16_t*)y, HW, C, total);
  else hipLaunchKernelGGL(easpp_fuse_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)br0, (const float*)br1, (const float*)br2, (const float*)br3, glob, w, (float*)y, HW, C, total);
  return check_launch("easpp_fuse");
}
extern "C" int spg_easpp_fuse_bwd(int dtype, const void* dy, const void* br0, const void* br1, const void* br2, const void* br3,
                                  const float* glob, const float* w, void* d0, void* d1, void* d2, void* d3, float* dglob,
                                  float* dw, int B, long HW, int C, float* red_ws, long red_ws_floats, unsigned* red_counter,
                                  spg_stream_t stream) {
  const long total = (long)B * HW * 4 * C;
  long rpb = cdiv(HW, 32);
  if (rpb < 8) rpb = 8;
  dim3 g2(cdiv(HW, rpb), B);
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= 32L * B * 6 * C, "easpp_fuse_bwd: needs 32*B*6*C floats of scratch and one zeroed counter");
  if (dtype == SPG_BF16) {
    hipLaunchKernelGGL(easpp_fuse_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, w, (bf16_t*)d0, (bf16_t*)d1, (bf16_t*)d2, (bf16_t*)d3, HW, C, total);
    hipLaunchKernelGGL(easpp_fuse_bwd_reduce_kernel<bf16_t>, g2, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)br0, (const bf16_t*)br1, (const bf16_t*)br2, (const bf16_t*)br3, glob, w, dglob, dw, HW, C, rpb, red_ws, red_counter);
  } else {
    hipLaunchKernelGGL(easpp_fuse_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)d0, (float*)d1, (float*)d2, (float*)d3, HW, C, total);
    hipLaunchKernelGGL(easpp_fuse_bwd_reduce_kernel<float>, g2, dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)br0, (const float*)br1, (const float*)br2, (const float*)br3, glob, w, dglob, dw, HW, C, rpb, red_ws, red_counter);
  }
  return check_launch("easpp_fuse_bwd");
}
extern "C" int spg_head1x1(int dtype, const void* x, const float* w, const float* b, void* y, long M, int C, spg_stream_t stream) {
  const int v = vec_of(dtype);
  const int lpp = C / v;
  SPG_REQUIRE(C % v == 0 && lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0, "head1x1: C=%d/%d must be a power of two <= 64", C, v);
  const int ppw = 64 / lpp;
  const int grid = ew_grid((M + ppw - 1) / ppw * 64);
  if (dtype == SPG_BF16) hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, b, (bf16_t*)y, M, C);
  else hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, b, (float*)y, M, C);
  return check_launch("head1x1");
}
static inline long head_bwd_blocks(int C) { long b = 32768 / C; return b < 64 ? 64 : (b > 512 ? 512 : b); }
extern "C" long spg_head1x1_bwd_workspace_floats(int C) { return head_bwd_blocks(C) * (C + 1); }
extern "C" int spg_head1x1_bwd(int dtype, const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, long M,
                               int C, int accumulate, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream) {
  const int v = vec_of(dtype);
  SPG_REQUIRE(C % v == 0 && C / v <= 256, "head1x1_bwd: C alignment");
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= spg_head1x1_bwd_workspace_floats(C), "head1x1_bwd: reduction workspace too small");
  const long rpar = 256 / (C / v);
  long rpb = cdiv(M, head_bwd_blocks(C));
  if (rpb < rpar * 8) rpb = rpar * 8;
  if (dtype == SPG_BF16) {
    hipLaunchKernelGGL(head_bwd_dx_kernel<bf16_t>, dim3(ew_grid(M * (C / v))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, w, (bf16_t*)dx, M, C, accumulate);
    hipLaunchKernelGGL(head_bwd_dw_kernel<bf16_t>, dim3(cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, dw, db, M, C, rpb, red_ws, red_counter);
  } else {
    hipLaunchKernelGGL(head_bwd_dx_kernel<float>, dim3(ew_grid(M * (C / v))), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, M, C, accumulate);
    hipLaunchKernelGGL(head_bwd_dw_kernel<float>, dim3(cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, dw, db, M, C, rpb, red_ws, red_counter);
  }
  return check_launch("head1x1_bwd");
}
