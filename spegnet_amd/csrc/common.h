// Shared device/host helpers for the SPEGNet gfx950 kernels.
// Storage type T is float (parity path, exact-f32 MFMA) or bf16 (fast path, bf16 MFMA, f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/spegnet_hip.h"

namespace spg {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int WAVE = 64;

// ---- error plumbing (no exceptions across the C ABI) -------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
#define SPG_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      spg::set_error(__VA_ARGS__);                \
      return SPG_ERR_BAD_ARG;                     \
    }                                             \
  } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, keeps NaN a NaN
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

template <typename T> struct ST;  // storage traits
template <> struct ST<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ST<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte chunk <-> floats
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& c, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& c, float* f) {
  f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& c, float* f) {
  f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
  f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
  f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
  f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
  u32x4 c; c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]); c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]);
  return c;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
  u32x4 c; c.x = pack2bf(f[0], f[1]); c.y = pack2bf(f[2], f[3]); c.z = pack2bf(f[4], f[5]); c.w = pack2bf(f[6], f[7]);
  return c;
}
template <typename T> __device__ __forceinline__ u32x4 ld16(const T* p) { return *reinterpret_cast<const u32x4*>(p); }
template <typename T> __device__ __forceinline__ void st16(T* p, const u32x4& c) { *reinterpret_cast<u32x4*>(p) = c; }

// ---- wave / block reductions ------------------------------------------------------------------
// (Six dependent ds_bpermute_b32 in the ISA.  A DPP form -- quad swaps, row rotations, four v_readlane -- was built and measured: LayerNorm
// forward 6.0 -> 5.7 us, backward 6.1 -> 6.3, the step level at 21.1-21.2 ms: not worth a different summation order everywhere.)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a block of NT threads; result valid in every thread. smem: >= NT/64 floats
template <int NT> __device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += smem[i];
  return r;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 rounding level): ~12 VALU ops + one v_exp instead of the
// ~40-instruction libm erff -- the exact-erf GELU of nn.GELU() to within the parity budget, at a fraction of the epilogue cost.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp): the IEEE sequence of __frcp_rn is ~10 VALU per element
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// gelu(x) and gelu'(x) together: the exponential inside erf_fast(x / sqrt 2) IS exp(-x^2 / 2), the density's -- the derivative costs four
// more operations when it is formed next to the value (forward epilogue), against a second erf + exp when it is re-derived from the saved
// pre-activation in the backward epilogue.  g is bit-identical to gelu_f(x).
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& d) {
  const float u = x * 0.70710678118654752440f;
  const float ax = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp): the IEEE sequence of __frcp_rn is ~10 VALU per element
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-ax * ax);
  const float erf = copysignf(1.f - poly * e, u);
  g = 0.5f * x * (1.f + erf);
  d = 0.5f * (1.f + erf) + x * (0.39894228040143267794f * e);
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }   // (v_rcp_f32, 1 ulp)

// ---- deterministic cross-workgroup reductions ----------------------------------------------------------
// Every reduction that spans workgroups writes one partial vector per workgroup and the workgroup that arrives LAST at a counter
// sums the partials in a fixed order (no float atomics: results are bitwise reproducible, which train-mode BatchNorm statistics
// need).  Hand-off = cdna_hip_programming.md Guideline 16, R1 in its counter form: partials are stored WRITE-THROUGH (st_part: an
// agent-scope relaxed store = global_store sc1, so no release fence and no L2 write-back per workgroup), every storing wave drains
// its stores, workgroup barrier, one lane draws a ticket; the last arriver acquires at agent scope (drops this CU's L1) before the
// workgroup reads the other workgroups' partials with plain loads.  The last arriver re-arms the counter (0), so a counter only
// has to be zero before its FIRST use.
__device__ __forceinline__ void st_part(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned expected, unsigned* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (t == expected - 1u) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0u;
}
// out[c] (+)= sum_{b < nblk} part[b * ld + c] for c < ncols, summed in the fixed order b = w, w+NW, ... per wave w, then over the
// waves in order (NT threads = NW waves, 64 consecutive columns per wave pass).  scratch: NT floats of LDS.
template <int NT>
__device__ __forceinline__ void finish_partials(const float* __restrict__ part, int nblk, int ld, int ncols, float* __restrict__ out,
                                                int accumulate, float* scratch) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c0 = 0; c0 < ncols; c0 += 64) {
    const int c = c0 + lane;
    float s = 0.f;
    if (c < ncols) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains keep the loads in flight; order stays fixed
      int b = w;
      for (; b + 3 * NW < nblk; b += 4 * NW) {
        s0 += part[(long)b * ld + c];
        s1 += part[(long)(b + NW) * ld + c];
        s2 += part[(long)(b + 2 * NW) * ld + c];
        s3 += part[(long)(b + 3 * NW) * ld + c];
      }
      for (; b < nblk; b += NW) s0 += part[(long)b * ld + c];
      s = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    scratch[threadIdx.x] = s;
    __syncthreads();
    if (w == 0 && c < ncols) {
      float t = scratch[lane];
#pragma unroll
      for (int i = 1; i < NW; ++i) t += scratch[i * 64 + lane];
      out[c] = accumulate ? out[c] + t : t;
    }
  }
}

// The same finish for whole partial ROWS of ldp floats (ldp % 4 == 0): res[0..ldp) (LDS) = sum over the nblk rows, fixed order.
// A lane reads 16 bytes of a row, so one wave-instruction covers 1 KiB of a row, and eight row loads stay in flight per lane:
// the last workgroup streams the partials instead of waiting out one memory latency per row (with 4-byte column loads the
// finish of 1024 x 256 partials cost more than the reduction it finishes).  scratch: NT * 4 floats of LDS; ends with a barrier.
template <int NT>
__device__ __forceinline__ void finish_rows(const float* __restrict__ part, int nblk, int ldp, float* res, float* scratch) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nq = ldp >> 2;
  for (int q0 = 0; q0 < nq; q0 += 64) {
    const int q = q0 + lane;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (q < nq) {
      const float* col = part + q * 4;
      int b = w;
      for (; b + 7 * NW < nblk; b += 8 * NW) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(col + (long)b * ldp);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(col + (long)(b + NW) * ldp);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(col + (long)(b + 2 * NW) * ldp);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(col + (long)(b + 3 * NW) * ldp);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(col + (long)(b + 4 * NW) * ldp);
        const f32x4 v5 = *reinterpret_cast<const f32x4*>(col + (long)(b + 5 * NW) * ldp);
        const f32x4 v6 = *reinterpret_cast<const f32x4*>(col + (long)(b + 6 * NW) * ldp);
        const f32x4 v7 = *reinterpret_cast<const f32x4*>(col + (long)(b + 7 * NW) * ldp);
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        a0 += v4; a1 += v5; a2 += v6; a3 += v7;
      }
      for (; b < nblk; b += NW) a0 += *reinterpret_cast<const f32x4*>(col + (long)b * ldp);
    }
    const f32x4 s = (a0 + a1) + (a2 + a3);
    __syncthreads();
    *reinterpret_cast<f32x4*>(scratch + threadIdx.x * 4) = s;
    __syncthreads();
    if (w == 0 && q < nq) {
      f32x4 t = *reinterpret_cast<const f32x4*>(scratch + lane * 4);
#pragma unroll
      for (int i = 1; i < NW; ++i) t += *reinterpret_cast<const f32x4*>(scratch + (i * 64 + lane) * 4);
      *reinterpret_cast<f32x4*>(res + q * 4) = t;
    }
  }
  __syncthreads();
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace spg
