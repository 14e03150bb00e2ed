// Optimizer step of Trainer._process_batch (reference engine/trainer.py:399-409, param groups :274-306) over a
// flat fp32 arena: global gradient L2 norm (clip_grad_norm_) + AdamW, with NO host synchronisation -- the clip
// coefficient, learning rates and the step counter are read from device memory so the whole train step can sit
// inside one hipGraph.  HBM-bound: p,g,m,v read + p,m,v written once per step (28 B / parameter).
#include "common.h"

namespace spg {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long n4,
                                                    float* __restrict__ part, unsigned* __restrict__ counter) {
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  __shared__ float red[4];
  __shared__ float scratch[256];
  __shared__ unsigned s_last;
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) st_part(part + blockIdx.x, s);
  if (!arrive_last(counter, gridDim.x, &s_last)) return;
  // fixed-order finish: thread t adds partials t, t+256, ...; then the 256 sums in index order
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += part[i];
  scratch[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256; ++i) tot += scratch[i];
    out[0] = tot;
  }
}

// Global gradient norm when part of the gradients' sums of squares already exists: out[0] = sum over `chunks` (disjoint ranges of x, each
// <= 16384 floats, a multiple of 4, one workgroup each) + sum over the `extras` arrays (per-block sums written by spg_gemm_tn_blocks'
// owners), all in a fixed order: per-chunk partials and every extra array are summed by the workgroup that arrives last, thread t
// taking elements t, t + 256, ... of the concatenated list, then the 256 sums in index order.  Deterministic.
struct SqChunk { long off; int n4; int pad; };
constexpr int SQ_MAX_EXTRAS = 32;
struct SqExtras { const float* ptr[SQ_MAX_EXTRAS]; int n[SQ_MAX_EXTRAS]; int count; };

__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ x, const SqChunk* __restrict__ chunks, SqExtras ex,
                                                         float* __restrict__ out, float* __restrict__ part, unsigned* __restrict__ counter) {
  const SqChunk c = chunks[blockIdx.x];
  const float* p = x + c.off;
  float s = 0.f;
  for (int i = threadIdx.x; i < c.n4; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + i * 4L);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  __shared__ float red[4];
  __shared__ float scratch[256];
  __shared__ unsigned s_last;
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) st_part(part + blockIdx.x, s);
  if (!arrive_last(counter, gridDim.x, &s_last)) return;
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += part[i];
  for (int e = 0; e < ex.count; ++e)
    for (int i = threadIdx.x; i < ex.n[e]; i += 256) t += ex.ptr[e][i];
  scratch[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256; ++i) tot += scratch[i];
    out[0] = tot;
  }
}

// One AdamW element update with the arithmetic pinned (explicit fused / unfused operations, so the plain and the fused-pack kernel
// -- and any future variant -- produce bit-identical parameters whatever the compiler's contraction choices around the call).
struct AdamC { float coef, b1, b2, eps, bc1, bc2s; };
__device__ __forceinline__ float adam1(float& p, float g, float& m, float& v, float lr_, float wd_, const AdamC& a) {
  const float gg = __fmul_rn(g, a.coef);
  p = __fmul_rn(p, __fsub_rn(1.f, __fmul_rn(lr_, wd_)));
  m = __fmaf_rn(a.b1, m, __fmul_rn(__fsub_rn(1.f, a.b1), gg));
  v = __fmaf_rn(a.b2, v, __fmul_rn(__fmul_rn(__fsub_rn(1.f, a.b2), gg), gg));
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), a.bc2s), a.eps);
  p = __fsub_rn(p, __fmul_rn(__fdiv_rn(lr_, a.bc1), __fdiv_rn(m, denom)));
  return p;
}

// group_of_chunk[i/256] selects (lr, wd); every parameter starts on a 256-element boundary of the arena.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const unsigned char* __restrict__ group_of_chunk,
                                                    const float* __restrict__ lr, const float* __restrict__ wd,
                                                    const float* __restrict__ gnorm_sq, const float* __restrict__ step_f,
                                                    float clip, float b1, float b2, float eps, float grad_scale, long n4,
                                                    int zero_grad) {
  AdamC a;
  a.coef = grad_scale;
  if (clip > 0.f) {
    const float tot = sqrtf(gnorm_sq[0]) * grad_scale;
    a.coef *= fminf(1.f, clip / (tot + 1e-6f));
  }
  const float t = step_f[0];
  a.b1 = b1; a.b2 = b2; a.eps = eps;
  a.bc1 = 1.f - powf(b1, t); a.bc2s = sqrtf(1.f - powf(b2, t));
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int grp = group_of_chunk[i >> 6];
    const float lr_ = lr[grp], wd_ = wd[grp];
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4), vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { float pe = pv[e], me = mv[e], ve = vv[e]; adam1(pe, gv[e], me, ve, lr_, wd_, a); pv[e] = pe; mv[e] = me; vv[e] = ve; }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
    if (zero_grad) *reinterpret_cast<f32x4*>(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};   // next step accumulates into a clean arena
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// AdamW fused with the weight re-pack: the update kernel already holds every new parameter value in registers, so it also writes
// the compute-dtype copies the GEMM kernels read next step ([N][K] and [K][N] of every Linear / 1x1 weight, [Co][tap][Ci] and
// [Ci][tap'][Co] of the 3x3 convolutions, the zero-padded patch-embed matrix, the bf16 qkv biases).  That replaces a second pass
// over the fp32 masters (pack_batch: 0.85 GB read + 0.85 GB written per step) by 4 more bytes written per matrix element here.
// Work items: 64 x 64 tiles of a matrix (transposed copy through LDS), 4096-element chunks of anything else.  A job table in
// device memory (built once per model by the host) maps items to parameters; a block walks a contiguous range of items.
// ------------------------------------------------------------------------------------------------------------------------------
struct OptJob {
  long off;        // arena offset of element (0, 0) of this job
  void* dst;       // compute-dtype copy: matrix [R][ldd] / flat [n] / conv forward pack; may be null
  void* dst_t;     // transposed copy [C][R] / conv dgrad pack; may be null
  int R, C;        // matrix: rows, columns; flat: R = 1, C = n; conv: R = Co, C = Ci
  int lds, ldd;    // matrix: source row stride (arena elements), dst row stride (elements)
  int item0;       // first work item of this job
  int kind;        // 0 flat, 1 matrix, 2 conv3x3 [Co][Ci][3][3]
};
enum { OPT_FLAT = 0, OPT_MATRIX = 1, OPT_CONV = 2,
       OPT_KEEP_G = 256 };   // flag on a matrix job's kind: its gradients are NOT cleared (the caller's next backward stores them whole:
                             // spg_gemm_tn_blocks with overwrite)

template <typename T>
__global__ __launch_bounds__(256) void adamw_pack_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, const unsigned char* __restrict__ group_of_chunk,
                                                         const float* __restrict__ lr, const float* __restrict__ wd,
                                                         const float* __restrict__ gnorm_sq, const float* __restrict__ step_f,
                                                         float clip, float b1, float b2, float eps, float grad_scale, int zero_grad,
                                                         const OptJob* __restrict__ jobs, int njobs, int total_items, int items_per_block) {
  __shared__ float tile[64][65];
  AdamC a;
  a.coef = grad_scale;
  if (clip > 0.f) {
    const float tot = sqrtf(gnorm_sq[0]) * grad_scale;
    a.coef *= fminf(1.f, clip / (tot + 1e-6f));
  }
  const float t = step_f[0];
  a.b1 = b1; a.b2 = b2; a.eps = eps;
  a.bc1 = 1.f - powf(b1, t); a.bc2s = sqrtf(1.f - powf(b2, t));
  int it = blockIdx.x * items_per_block;
  const int it_end = min(total_items, it + items_per_block);
  if (it >= it_end) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {  // last job with item0 <= it
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].item0 <= it) lo = mid; else hi = mid - 1;
  }
  int j = lo;
  OptJob jb = jobs[j];
  int next0 = j + 1 < njobs ? jobs[j + 1].item0 : total_items;
  const int tid = threadIdx.x;
  for (; it < it_end; ++it) {
    while (it >= next0) { ++j; jb = jobs[j]; next0 = j + 1 < njobs ? jobs[j + 1].item0 : total_items; }
    const int ti = it - jb.item0;
    T* dst = reinterpret_cast<T*>(jb.dst);
    T* dst_t = reinterpret_cast<T*>(jb.dst_t);
    if ((jb.kind & 255) == OPT_MATRIX) {
      const bool clear_g = zero_grad != 0 && (jb.kind & OPT_KEEP_G) == 0;
      const int R = jb.R, C = jb.C;
      const int tiles_c = (C + 63) >> 6;
      const int r0 = (ti / tiles_c) * 64, c0 = (ti % tiles_c) * 64;
      const bool vec = ((jb.lds | C) & 3) == 0;        // rows start on 16-byte boundaries (parameters start on 256-element ones)
      if (dst_t) __syncthreads();                       // the previous item's transposed reads of tile[][]
      if (vec) {
        const int cq = tid & 15, rr = tid >> 4;
        const int c = c0 + 4 * cq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = r0 + rr + 16 * k;
          if (r < R && c < C) {
            const long i = jb.off + (long)r * jb.lds + c;
            const int grp = group_of_chunk[i >> 8];
            const float lr_ = lr[grp], wd_ = wd[grp];
            // (streamed once per step, 3.4 GB in and out: non-temporal so they do not displace the packed weights in L2 / MALL)
            f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + i));
            const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
            f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i)), vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pe = pv[e], me = mv[e], ve = vv[e]; adam1(pe, gv[e], me, ve, lr_, wd_, a); pv[e] = pe; mv[e] = me; vv[e] = ve; }
            __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p + i));
            __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + i));
            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
            if (clear_g) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(g + i));
            if (dst) {
              T* d = dst + (long)r * jb.ldd + c;
              if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x2*>(d) = u32x2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
              else *reinterpret_cast<f32x4*>(d) = pv;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[rr + 16 * k][4 * cq + e] = pv[e];
          }
        }
      } else {
        for (int e = tid; e < 64 * 64; e += 256) {
          const int rl = e >> 6, cl = e & 63, r = r0 + rl, c = c0 + cl;
          if (r < R && c < C) {
            const long i = jb.off + (long)r * jb.lds + c;
            const int grp = group_of_chunk[i >> 8];
            float pv = p[i], mv = m[i], vv = v[i];
            adam1(pv, g[i], mv, vv, lr[grp], wd[grp], a);
            p[i] = pv; m[i] = mv; v[i] = vv;
            if (clear_g) g[i] = 0.f;
            if (dst) ST<T>::st(dst + (long)r * jb.ldd + c, pv);
            tile[rl][cl] = pv;
          }
        }
      }
      if (dst_t) {
        __syncthreads();
        // dst_t[c][r]: a thread writes 16 consecutive r of one c
        const int cl = tid >> 2, rq = (tid & 3) * 16, c = c0 + cl;
        if (c < C) {
          T* d = dst_t + (long)c * R + r0 + rq;
          if (sizeof(T) == 2 && (R & 7) == 0 && r0 + rq + 16 <= R) {
            float f0[8], f1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { f0[e] = tile[rq + e][cl]; f1[e] = tile[rq + 8 + e][cl]; }
            st16(d, pack16<T>(f0));
            st16(d + 8, pack16<T>(f1));
          } else {
            for (int e = 0; e < 16; ++e)
              if (r0 + rq + e < R) ST<T>::st(d + e, tile[rq + e][cl]);
          }
        }
      }
    } else {
      // flat / conv: 4096 elements per item, 4 x float4 per thread.  Updating past n up to the next multiple of 4 is harmless:
      // parameters are padded to 256-element boundaries with zeros, whose update is again zero.
      const long n = jb.kind == OPT_CONV ? (long)jb.R * jb.C * 9 : (long)jb.C;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long e0 = (long)ti * 4096 + k * 1024 + tid * 4;
        if (e0 < n) {
          const long i = jb.off + e0;
          const int grp = group_of_chunk[i >> 8];
          const float lr_ = lr[grp], wd_ = wd[grp];
          f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
          const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
          f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
          for (int e = 0; e < 4; ++e) { float pe = pv[e], me = mv[e], ve = vv[e]; adam1(pe, gv[e], me, ve, lr_, wd_, a); pv[e] = pe; mv[e] = me; vv[e] = ve; }
          *reinterpret_cast<f32x4*>(p + i) = pv;
          *reinterpret_cast<f32x4*>(m + i) = mv;
          *reinterpret_cast<f32x4*>(v + i) = vv;
          if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
          if (jb.kind == OPT_FLAT) {
            if (dst) {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (e0 + e < n) ST<T>::st(dst + e0 + e, pv[e]);
            }
          } else {
            const int Ci = jb.C, Co = jb.R;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const long q = e0 + e;
              if (q < n) {
                const int tap = (int)(q % 9);
                const long tt = q / 9;
                const int ci = (int)(tt % Ci), co = (int)(tt / Ci);
                if (dst) ST<T>::st(dst + ((long)co * 9 + tap) * Ci + ci, pv[e]);
                if (dst_t) ST<T>::st(dst_t + ((long)ci * 9 + (8 - tap)) * Co + co, pv[e]);
              }
            }
          }
        }
      }
    }
  }
}

__global__ void add_scalar_kernel(float* x, float a) { x[0] += a; }

}  // namespace spg

using namespace spg;

extern "C" int spg_sumsq(const float* x, float* out, long n, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream) {
  SPG_REQUIRE(n % 4 == 0, "sumsq: n=%ld must be a multiple of 4", n);
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= 2048, "sumsq: needs 2048 floats of scratch and one zeroed counter");
  long g = (n / 4 + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, out, n / 4, red_ws, red_counter);
  return check_launch("sumsq");
}

extern "C" int spg_sumsq_fold(const float* x, const void* chunks, int nchunks, int nextras, const float* const* extra_ptr, const int* extra_n,
                              float* out, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream) {
  SPG_REQUIRE(x && chunks && nchunks >= 1 && out, "sumsq_fold: needs at least one chunk of the gradient arena");
  SPG_REQUIRE(nextras >= 0 && nextras <= SQ_MAX_EXTRAS && (nextras == 0 || (extra_ptr && extra_n)), "sumsq_fold: 0..%d extra arrays, got %d", SQ_MAX_EXTRAS, nextras);
  SPG_REQUIRE(red_ws && red_counter && red_ws_floats >= nchunks, "sumsq_fold: needs %d floats of scratch and one zeroed counter", nchunks);
  static_assert(sizeof(SqChunk) == 16, "SqChunk layout is part of the ABI");
  SqExtras ex;
  ex.count = nextras;
  for (int i = 0; i < SQ_MAX_EXTRAS; ++i) { ex.ptr[i] = i < nextras ? extra_ptr[i] : nullptr; ex.n[i] = i < nextras ? extra_n[i] : 0; }
  for (int i = 0; i < nextras; ++i) SPG_REQUIRE(ex.ptr[i] && ex.n[i] > 0, "sumsq_fold: extra array %d is empty", i);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, x, (const SqChunk*)chunks, ex, out, red_ws, red_counter);
  return check_launch("sumsq_fold");
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

/* AdamW + weight re-pack in one pass (see adamw_pack_kernel).  jobs: DEVICE array of njobs records
 *   struct { long off; void* dst; void* dst_t; int R, C, lds, ldd, item0, kind; }   (48 bytes, kind 0 flat / 1 matrix / 2 conv3x3;
 *   kind | 256 on a matrix job: zero_grad leaves that job's gradients as they are -- the caller's next backward overwrites them)
 * that together cover every parameter of the arena exactly once; total_items = sum of the jobs' work items (matrix: 64 x 64 tiles,
 * flat / conv: 4096-element chunks).  dtype is the compute dtype of the copies.                                                    */
extern "C" int spg_adamw_pack(int dtype, float* p, float* g, float* m, float* v, const unsigned char* group_of_chunk, const float* lr,
                              const float* wd, const float* gnorm_sq, float* step_f, float clip, float beta1, float beta2, float eps,
                              float grad_scale, int zero_grad, const void* jobs, int njobs, int total_items, spg_stream_t stream) {
  SPG_REQUIRE(njobs > 0 && total_items > 0 && jobs, "adamw_pack: empty job table");
  static_assert(sizeof(OptJob) == 48, "OptJob layout is part of the ABI");
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_f, 1.0f);
#ifndef SPG_ADAMW_IPB
#define SPG_ADAMW_IPB 1   // work items per block (8: 1.72 ms for sumsq + AdamW + re-pack in the step, 2: 1.68, 1: 1.65 -- more blocks in flight)
#endif
  const int ipb = SPG_ADAMW_IPB;
  const int grid = (total_items + ipb - 1) / ipb;
  if (dtype == SPG_BF16)
    hipLaunchKernelGGL(adamw_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, group_of_chunk, lr, wd, gnorm_sq,
                       step_f, clip, beta1, beta2, eps, grad_scale, zero_grad, (const OptJob*)jobs, njobs, total_items, ipb);
  else
    hipLaunchKernelGGL(adamw_pack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, group_of_chunk, lr, wd, gnorm_sq,
                       step_f, clip, beta1, beta2, eps, grad_scale, zero_grad, (const OptJob*)jobs, njobs, total_items, ipb);
  return check_launch("adamw_pack");
}
