// Optimizer step of Trainer._process_batch (reference engine/trainer.py:399-409, param groups :274-306) over a
// flat fp32 arena: global gradient L2 norm (clip_grad_norm_) + AdamW, with NO host synchronisation -- the clip
// coefficient, learning rates and the step counter are read from device memory so the whole train step can sit
// inside one hipGraph.  HBM-bound: p,g,m,v read + p,m,v written once per step (28 B / parameter).
#include "common.h"

namespace spg {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long n4) {
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  __shared__ float red[4];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// group_of_chunk[i/256] selects (lr, wd); every parameter starts on a 256-element boundary of the arena.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const unsigned char* __restrict__ group_of_chunk,
                                                    const float* __restrict__ lr, const float* __restrict__ wd,
                                                    const float* __restrict__ gnorm_sq, const float* __restrict__ step_f,
                                                    float clip, float b1, float b2, float eps, float grad_scale, long n4,
                                                    int zero_grad) {
  float coef = grad_scale;
  if (clip > 0.f) {
    const float tot = sqrtf(gnorm_sq[0]) * grad_scale;
    coef *= fminf(1.f, clip / (tot + 1e-6f));
  }
  const float t = step_f[0];
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int grp = group_of_chunk[i >> 6];
    const float lr_ = lr[grp], wd_ = wd[grp];
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4), vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * coef;
      pv[e] *= 1.f - lr_ * wd_;
      mv[e] = b1 * mv[e] + (1.f - b1) * gg;
      vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[e]) / bc2s + eps;
      pv[e] -= (lr_ / bc1) * (mv[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
    if (zero_grad) *reinterpret_cast<f32x4*>(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};   // next step accumulates into a clean arena
  }
}

__global__ void add_scalar_kernel(float* x, float a) { x[0] += a; }

}  // namespace spg

using namespace spg;

extern "C" int spg_sumsq(const float* x, float* out, long n, spg_stream_t stream) {
  SPG_REQUIRE(n % 4 == 0, "sumsq: n=%ld must be a multiple of 4", n);
  long g = (n / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, out, n / 4);
  return check_launch("sumsq");
}

extern "C" int spg_adamw(float* p, float* g, float* m, float* v, const unsigned char* group_of_chunk, const float* lr,
                         const float* wd, const float* gnorm_sq, float* step_f, float clip, float beta1, float beta2, float eps,
                         float grad_scale, int zero_grad, long n, spg_stream_t stream) {
  SPG_REQUIRE(n % 256 == 0, "adamw: arena size %ld must be a multiple of 256", n);
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_f, 1.0f);
  long gr = (n / 4 + 255) / 256;
  if (gr > 4096) gr = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)gr), dim3(256), 0, (hipStream_t)stream, p, g, m, v, group_of_chunk, lr, wd, gnorm_sq,
                     step_f, clip, beta1, beta2, eps, grad_scale, n / 4, zero_grad);
  return check_launch("adamw");
}
