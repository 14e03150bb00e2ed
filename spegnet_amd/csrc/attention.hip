// Windowed multi-head attention for the Hiera trunk on gfx950 MFMA (forward, dQ, dK/dV).
//
// Follows sam2 Hiera's MultiScaleAttention (reference call site models/feature_encoding.py:236; algorithm in
// SURVEY.md §8 row E): the block's tokens are LayerNorm-ed, window-partitioned WITH zero padding, projected by
// the qkv Linear, q optionally 2x2 max-pooled, softmax(q k^T / sqrt(hd)) v per window.  Here nothing is
// partitioned or padded in memory: a workgroup enumerates the VALID tokens of its window's rectangle straight
// from the [B,H,W,3,heads,hd] projection output; the padded slots of the reference (whose k and v equal the
// qkv bias because LN output is zero there) become ONE virtual key of multiplicity n_pad, i.e. an additive
// log(n_pad) on its score.  Masked tile slots get -1e30.  This is the same sum, re-ordered.
//
// One tile engine serves the three kernels.  A wave keeps 16 "stationary" rows (queries, or keys in the dK/dV
// kernel) as MFMA B operands in registers; tiles of 64 "streamed" tokens sit in LDS as ONE row image [token][d]: it
// is the A operand of the score-type products as is, and the A operand of the products that sum over tokens through the
// hardware transpose read ds_read_b64_tr_b16 (bf16) / strided reads (f32).  Scores are computed transposed (streamed token on the accumulator row, stationary row on
// the lane) so the softmax is lane-local and P / dS feed the next MFMA from registers with a permuted k order
// (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace spg {

constexpr float NEG_BIG = -1.0e30f;
constexpr int AT = 256;  // threads

struct AttnP {
  const void* qkv; const void* qp; const void* bias;  // T
  void* out; float* lse;
  const void* dout; void* dqkv; void* dqp; float* dbias; float* delta;
  int B, H, W, heads, ws, nwy, nwx, Hq, Wq, wsq, C;
  int wsx, wsxq;   // window WIDTH in key / query tokens (= ws / wsq except for packed small windows, below)
  int sub;         // > 0: a 'window' is `wsx / sub` real windows of width `sub` side by side; tokens attend only inside their own
  float scale;
};

struct Win {
  int b, y0, x0, hv, wv, nvalid, npad, y0q, x0q, wvq, nq;
  int krow0, qrow0;         // global row of the window's first key / query token
  float inv_wv, inv_wvq;    // 1 / window width (keys, queries): token -> (line, column) without an integer division
};
// c / d for 0 <= c < 2^16 and a small divisor given as inv = 1 / d (hardware reciprocal): (c + 0.5) / d sits at least 0.5 / d away from
// an integer, far more than the rounding of the two float operations.  An integer division by a run-time divisor is ~40 VALU
// instructions, and the resident kernels' staging did one per 16-byte chunk: address arithmetic, not the loads, was most of that phase
// (in-kernel stamps, round 2: 7700 of 9800 cycles before the loads were even issued).
// (Also measured: the eight dependent integer divisions of res_unit / get_win at the head of every workgroup -- ~3000 stamped cycles --
// replaced the same way, and a shift-and-mask thread -> (row, chunk) staging map: no change in launch time, 16.3 us either way.)
__device__ __forceinline__ int fast_div(int c, float inv) { return (int)(((float)c + 0.5f) * inv); }

__device__ __forceinline__ Win get_win(const AttnP& p, int widx) {
  Win w;
  const int per = p.nwy * p.nwx;
  w.b = widx / per;
  const int r = widx - w.b * per;
  const int wy = r / p.nwx, wx = r - wy * p.nwx;
  w.y0 = wy * p.ws; w.x0 = wx * p.wsx;
  w.hv = min(p.ws, p.H - w.y0); w.wv = min(p.wsx, p.W - w.x0);
  w.nvalid = w.hv * w.wv;
  w.npad = p.ws * p.wsx - w.nvalid;
  if (p.qp) {
    w.y0q = wy * p.wsq; w.x0q = wx * p.wsxq;
    w.wvq = w.wv >> 1;
    w.nq = (w.hv >> 1) * w.wvq;
  } else {
    w.y0q = w.y0; w.x0q = w.x0; w.wvq = w.wv; w.nq = w.nvalid;
  }
  w.krow0 = (w.b * p.H + w.y0) * p.W + w.x0;
  w.qrow0 = (w.b * p.Hq + w.y0q) * p.Wq + w.x0q;
  w.inv_wv = __builtin_amdgcn_rcpf((float)w.wv);
  w.inv_wvq = __builtin_amdgcn_rcpf((float)max(w.wvq, 1));
  return w;
}
// global row index (into [B,H,W]) of key token c (< nvalid) / (into [B,Hq,Wq]) of query token i (< nq)
// (the launcher checks B * H * W < 2^24 rows and < 2^32 qkv elements: 24-bit multiplies, 32-bit element offsets)
__device__ __forceinline__ long key_row(const AttnP& p, const Win& w, int c) {
  const int ly = fast_div(c, w.inv_wv), lx = c - __mul24(ly, w.wv);
  return w.krow0 + __mul24(ly, p.W) + lx;
}
__device__ __forceinline__ long q_row(const AttnP& p, const Win& w, int i) {
  const int ly = fast_div(i, w.inv_wvq), lx = i - __mul24(ly, w.wvq);
  return w.qrow0 + __mul24(ly, p.Wq) + lx;
}
// element offset of a token's row in qkv [rows, 3C] / in a [rows, C] tensor
__device__ __forceinline__ size_t row3c(const AttnP& p, long row) { return (size_t)__umul24((unsigned)row, (unsigned)(3 * p.C)); }
__device__ __forceinline__ size_t row1c(const AttnP& p, long row) { return (size_t)__umul24((unsigned)row, (unsigned)p.C); }

// Packed small windows (sub > 0: 4 x 4 windows of stage 2, four side by side in one 4 x 16 'window' = one 64-token tile, so that a
// workgroup's four waves and the 64-slot tile are all used instead of a quarter of each): a key / query pair interacts only if both lie in
// the same real window.  With 16-wide rows the real window of key c is (c % 16) / 4 and of query i is (i % 16) / 4 -- or (i % 8) / 2
// for 2 x 2-pooled queries -- so in the tile layouts below the test is a lane constant or depends on the register index only.
__device__ __forceinline__ bool same_sub(const AttnP& p, int qi, int kc) {
  const int ks = (kc & 15) >> 2;
  const int qs = p.qp ? ((qi & 7) >> 1) : ((qi & 15) >> 2);
  return ks == qs;
}

// Work units of the resident-window kernels, largest first (longest-processing-time order: the hardware dispatches workgroups in
// index order, one per CU): window classes A = full ws x ws windows, B = right-edge, C = bottom-edge, D = corner windows of an image,
// each window cut into `parts` runs of 8 row blocks (128 queries / keys) so that a full 16 x 16 window is two workgroups and not one
// that runs twice as long as the CUs holding the partial windows of the same image.  cnt = windows of the class per image.
struct ResPlan {
  int cnt[4], parts[4], first[4];   // first[c]: first unit of class c (all images' units of a class are consecutive)
  int total;
};
__device__ __forceinline__ void res_unit(const ResPlan& pl, const AttnP& p, int u, int& widx, int& part, int& nparts) {
  int c = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (u >= pl.first[i]) c = i;
  const int v = u - pl.first[c];
  nparts = pl.parts[c];
  part = v % nparts;
  const int wi = v / nparts;                 // window of class c, image-major
  const int b = wi / pl.cnt[c], k = wi - b * pl.cnt[c];
  const int nfy = p.H / p.ws, nfx = p.W / p.wsx;
  int wy, wx;
  if (c == 0) { wy = k / nfx; wx = k - wy * nfx; }
  else if (c == 1) { wy = k; wx = nfx; }     // right edge: one per full window row
  else if (c == 2) { wy = nfy; wx = k; }     // bottom edge: one per full window column
  else { wy = nfy; wx = nfx; }
  widx = (b * p.nwy + wy) * p.nwx + wx;
}
// (Measured and dropped, round 2: running an image's corner window as a second pass of its bottom-edge unit -- 256 workgroups, one round,
// instead of 320 -- left the forward at 16.9 us and cost the merged backward 4 us (the folded units became the longest and sat late in
// the order): a launch lasts as long as its longest workgroup, a full window's half, plus ~5 us; the short second-round units are free.)
static ResPlan res_plan(const AttnP& p, bool keys) {
  ResPlan pl;
  const int nfy = p.H / p.ws, nfx = p.W / p.wsx, ry = p.H % p.ws, rx = p.W % p.wsx;
  const int hv[4] = {p.ws, p.ws, ry, ry}, wv[4] = {p.wsx, rx, p.wsx, rx};
  const int cnt[4] = {nfy * nfx, rx ? nfy : 0, ry ? nfx : 0, (rx && ry) ? 1 : 0};
  int u = 0;
  for (int c = 0; c < 4; ++c) {
    const int nv = hv[c] * wv[c];
    const int rows = keys ? nv : (p.qp ? (hv[c] / 2) * (wv[c] / 2) : nv);    // (the virtual pad key rides with the last key block)
    int parts = (rows + 127) / 128;
    if (parts < 1) parts = 1;
    if (parts > 2) parts = 2;
    pl.cnt[c] = cnt[c]; pl.parts[c] = parts; pl.first[c] = u;
    u += p.B * cnt[c] * parts;
  }
  pl.total = u;
  return pl;
}

template <typename T, int HD> struct AC {  // attention constants
  static constexpr int VEC = ST<T>::VEC;
  static constexpr int NCH = HD / VEC;                               // 16-byte chunks per head row
  static constexpr int DB = (HD + 15) / 16;                          // 16-row d blocks
  static constexpr int KS = sizeof(T) == 2 ? (HD + 31) / 32 : HD / 4;  // k-steps over d
  static constexpr int DROW = sizeof(T) == 2 ? KS * 32 : DB * 16;    // padded d per row image row (>= 16*DB, zero filled)
  static constexpr int RS = DROW * (int)sizeof(T) + 16;              // row image stride (bytes)
  static constexpr int ROW_BYTES = 64 * RS;
  using Frag = typename std::conditional<sizeof(T) == 2, bf16x8_t, float>::type;
};

// ---- fragments ------------------------------------------------------------------------------------
// B operand of a score-type product, straight from global: row pointer (or null), k-step s.
template <typename T, int HD>
__device__ __forceinline__ typename AC<T, HD>::Frag load_row_frag_global(const T* row, int s, int q) {
  if constexpr (sizeof(T) == 2) {
    const int ch = 4 * s + q;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row && ch < AC<T, HD>::NCH) v = ld16(row + ch * 8);
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    return row ? row[4 * s + q] : 0.f;
  }
}
// A operand of a score-type product from a row image: token row `r`, k-step s.
template <typename T, int HD>
__device__ __forceinline__ typename AC<T, HD>::Frag load_row_frag_lds(const char* img, int r, int s, int q) {
  if constexpr (sizeof(T) == 2) {
    return *reinterpret_cast<const bf16x8_t*>(img + r * AC<T, HD>::RS + (32 * s + 8 * q) * 2);
  } else {
    return *reinterpret_cast<const float*>(img + r * AC<T, HD>::RS + (4 * s + q) * 4);
  }
}

// dot product of two row fragments (the lane's slice of a row); f32 accumulation
__device__ __forceinline__ float frag_dot(bf16x8_t a, bf16x8_t b, float acc) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(bf2{a[2 * e], a[2 * e + 1]}, bf2{b[2 * e], b[2 * e + 1]}, acc, false);
  return acc;
}
__device__ __forceinline__ float frag_dot(float a, float b, float acc) { return acc + a * b; }

template <typename T> struct MM;
template <> struct MM<bf16_t> {
  __device__ static __forceinline__ f32x4 mma(bf16x8_t a, bf16x8_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MM<float> {
  __device__ static __forceinline__ f32x4 mma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// acc[db] += sum over the 64 tile tokens of Img[token][d = 16db + lane&15] * vals[token][col = lane&15], with vals given as
// accumulator-layout registers pv[nb][r] (token = 16nb + 4q + r).  The A operand (rows d, k = tokens) is read from the
// ROW image [token][d]: bf16 through the hardware transpose read ds_read_b64_tr_b16 (a 16-lane group fetches 4 token rows x
// 16 d columns and each lane receives one column), f32 through plain strided reads.  No transposed copy of the tile exists.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
// NB < 4: only the tile's first NB 16-token blocks count (pv[nb >= NB] is not read; bf16 rounds NB up to a 32-token step, the odd block's
// operand half zero -- the image rows behind it must be finite, which staged zero rows are).
template <typename T, int HD, int NB = 4>
__device__ __forceinline__ void mma_over_tokens(const char* rimg, const float (&pv)[4][4], int lane,
                                                f32x4 (&acc)[AC<T, HD>::DB]) {
  constexpr int DB = AC<T, HD>::DB, RS = AC<T, HD>::RS;
  const int r15 = lane & 15, q = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    const int qq = r15 >> 2, p = r15 & 3;   // lane 4qq+p of its 16-lane group addresses token row qq, d columns 4p..4p+3
#pragma unroll
    for (int s = 0; s < (NB + 1) / 2; ++s) {
      u32x4 b;
      if constexpr (NB == 4)
        b = u32x4{pack2bf(pv[2 * s][0], pv[2 * s][1]), pack2bf(pv[2 * s][2], pv[2 * s][3]),
                  pack2bf(pv[2 * s + 1][0], pv[2 * s + 1][1]), pack2bf(pv[2 * s + 1][2], pv[2 * s + 1][3])};
      else
        b = u32x4{pack2bf(pv[2 * s][0], pv[2 * s][1]), pack2bf(pv[2 * s][2], pv[2 * s][3]),
                  2 * s + 1 < NB ? pack2bf(pv[2 * s + 1][0], pv[2 * s + 1][1]) : 0u, 2 * s + 1 < NB ? pack2bf(pv[2 * s + 1][2], pv[2 * s + 1][3]) : 0u};
      const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, b);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const char* a0 = rimg + (32 * s + 4 * q + qq) * RS + (16 * db + 4 * p) * 2;
        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a0);
        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 16 * RS));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t a = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        acc[db] = MM<T>::mma(__builtin_bit_cast(bf16x8_t, a), bf, acc[db]);
      }
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const float a = *reinterpret_cast<const float*>(rimg + (16 * nb + 4 * q + r) * RS + (16 * db + r15) * 4);
          acc[db] = MM<T>::mma(a, pv[nb][r], acc[db]);
        }
  }
}

// score-type product for the whole tile: acc[nb] = sum_d Rimg[token 16nb + lane&15][d] * frag[d][col]
template <typename T, int HD, int NB = 4>
__device__ __forceinline__ void mma_scores(const char* rimg, const typename AC<T, HD>::Frag (&bf)[AC<T, HD>::KS], int lane,
                                           f32x4 (&acc)[4]) {
  const int r15 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < AC<T, HD>::KS; ++s)
      acc[nb] = MM<T>::mma(load_row_frag_lds<T, HD>(rimg, nb * 16 + r15, s, q), bf[s], acc[nb]);
  }
}

// ---- staging: 64 token rows (pointers in LDS, null = zero row) -> row image [token][d] ----------
template <typename T, int HD>
__device__ __forceinline__ void stage_tile(const T* const* ptrs, int off, char* rimg) {
  constexpr int VEC = AC<T, HD>::VEC, NCH = AC<T, HD>::NCH, RS = AC<T, HD>::RS;
  for (int p = threadIdx.x; p < 64 * NCH; p += AT) {
    const int ch = p % NCH, t = p / NCH;
    const T* s = ptrs[t];
    const u32x4 v = s ? ld16(s + off + ch * VEC) : u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(rimg + t * RS + ch * 16) = v;
  }
}

// the same for NT tiles at once: every load of a thread is requested before the first LDS write (stage_tile's pointer-read -> load ->
// write chain is one memory round trip per item: 3 per tile at head dim 72, and the tiles came one after the other -- 12 dependent round
// trips in the one-pass small-window backward).  A null row reads a fixed valid address (`safe`) and is replaced by zeros.
#ifndef SPG_DQ_STAGE_TILES      // (tools/ A/B builds) the tiled dQ kernel: 52.7 -> 38.6 us per global block with the batched form, 12 more VGPRs notwithstanding
#define SPG_DQ_STAGE_TILES 1
#endif
#ifndef SPG_DKV_STAGE_TILES
#define SPG_DKV_STAGE_TILES 0
#endif
struct StageSrc { const void* const* ptrs; int off; char* img; };
template <typename T, int HD, int NT>
__device__ __forceinline__ void stage_tiles(const StageSrc (&src)[NT], const T* safe) {
  constexpr int VEC = AC<T, HD>::VEC, NCH = AC<T, HD>::NCH, RS = AC<T, HD>::RS;
  constexpr int NIT = (64 * NCH + AT - 1) / AT;
  u32x4 v[NT][NIT];
  bool nz[NT][NIT];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = threadIdx.x + it * AT;
      const int pc = p < 64 * NCH ? p : 0;
      const int ch = pc % NCH, t = pc / NCH;
      const T* s = reinterpret_cast<const T* const*>(src[k].ptrs)[t];
      nz[k][it] = s != nullptr;
      v[k][it] = ld16((s ? s + src[k].off : safe) + ch * VEC);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = threadIdx.x + it * AT;
      if (p < 64 * NCH) {
        const int ch = p % NCH, t = p / NCH;
        *reinterpret_cast<u32x4*>(src[k].img + t * RS + ch * 16) = nz[k][it] ? v[k][it] : u32x4{0u, 0u, 0u, 0u};
      }
    }
}

// zero the parts of the images staging never writes (pad d columns / pad d rows)
template <typename T, int HD>
__device__ __forceinline__ void zero_images(char* base, int bytes) {
  for (int i = threadIdx.x * 16; i < bytes; i += AT * 16) *reinterpret_cast<u32x4*>(base + i) = u32x4{0u, 0u, 0u, 0u};
}

// store accumulators acc[db] (rows d = 16db + 4q + r, col = lane&15) * mul into dst row (4 consecutive d per reg group)
template <typename T, int HD>
__device__ __forceinline__ void store_rows_T(T* row, const f32x4 (&acc)[AC<T, HD>::DB], float mul, int lane) {
  const int q = lane >> 4;
#pragma unroll
  for (int db = 0; db < AC<T, HD>::DB; ++db) {
    const int d = db * 16 + q * 4;
    if (d < HD) {
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<u32x2*>(row + d) = u32x2{pack2bf(acc[db][0] * mul, acc[db][1] * mul), pack2bf(acc[db][2] * mul, acc[db][3] * mul)};
      } else {
        *reinterpret_cast<f32x4*>(row + d) = acc[db] * mul;
      }
    }
  }
}

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int HD, bool SUB = false>   // SUB: packed 4 x 4 windows (block mask same_sub)
__global__ __launch_bounds__(AT) void attn_fwd_kernel(AttnP p) {
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = kimg + A::ROW_BYTES;
  const T** kptr = reinterpret_cast<const T**>(vimg + A::ROW_BYTES);
  float* kb = reinterpret_cast<float*>(kptr + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const int head = blockIdx.y;
  const Win w = get_win(p, blockIdx.z);
  const int qi = blockIdx.x * 64 + wave * 16 + r15;
  if (blockIdx.x * 64 >= w.nq) return;  // uniform
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const bool qvalid = qi < w.nq;
  const long qrow = q_row(p, w, qvalid ? qi : 0);
  const T* qptr = qp ? qp + qrow * p.C + head * HD : qkv + qrow * 3 * p.C + head * HD;

  zero_images<T, HD>(smem, 2 * A::ROW_BYTES);
  typename A::Frag qf[A::KS];
#pragma unroll
  for (int s = 0; s < A::KS; ++s) qf[s] = load_row_frag_global<T, HD>(qptr, s, q);

  f32x4 o[A::DB];
#pragma unroll
  for (int db = 0; db < A::DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_BIG, l = 0.f;
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);
  const int ntiles = (nkeys + 63) >> 6;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // previous tile's readers done (also covers zero_images)
    if (tid < 64) {
      const int c = t * 64 + tid;
      const T* kp = nullptr;
      float b = NEG_BIG;
      if (c < w.nvalid) { kp = qkv + key_row(p, w, c) * 3 * p.C + p.C + head * HD; b = 0.f; }
      else if (c == w.nvalid && w.npad > 0) { kp = reinterpret_cast<const T*>(p.bias) + p.C + head * HD; b = __logf((float)w.npad); }
      kptr[tid] = kp; kb[tid] = b;
    }
    __syncthreads();
    {
      const StageSrc kv[2] = {{reinterpret_cast<const void* const*>(kptr), 0, kimg}, {reinterpret_cast<const void* const*>(kptr), p.C, vimg}};
      stage_tiles<T, HD, 2>(kv, qkv);
    }
    __syncthreads();
    f32x4 sacc[4];
    mma_scores<T, HD>(kimg, qf, lane, sacc);
    float pv[4][4];
    float mx = NEG_BIG;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + nb * 16 + q * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[nb][r] = sacc[nb][r] * p.scale + b4[r];
        if constexpr (SUB) { if (!same_sub(p, qi, t * 64 + nb * 16 + q * 4 + r)) pv[nb][r] = NEG_BIG; }
        mx = fmaxf(mx, pv[nb][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { pv[nb][r] = __expf(pv[nb][r] - mn); ps += pv[nb][r]; }
    l = l * alpha + ps;
#pragma unroll
    for (int db = 0; db < A::DB; ++db) o[db] *= alpha;
    mma_over_tokens<T, HD>(vimg, pv, lane, o);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (qvalid) {
    T* orow = reinterpret_cast<T*>(p.out) + qrow * p.C + head * HD;
    store_rows_T<T, HD>(orow, o, 1.f / l, lane);
    if (q == 0) p.lse[qrow * p.heads + head] = m + __logf(l);
  }
}

// =====================================================================================================
// backward, query side: dQ (+ delta = rowsum(dO*O) written for the dK/dV kernel)
// =====================================================================================================
template <typename T, int HD, bool SUB = false>   // SUB: packed 4 x 4 windows (block mask same_sub)
__global__ __launch_bounds__(AT) void attn_bwd_dq_kernel(AttnP p) {
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = kimg + A::ROW_BYTES;
  const T** kptr = reinterpret_cast<const T**>(vimg + A::ROW_BYTES);
  float* kb = reinterpret_cast<float*>(kptr + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const int head = blockIdx.y;
  const Win w = get_win(p, blockIdx.z);
  if (blockIdx.x * 64 >= w.nq) return;
  const int qi = blockIdx.x * 64 + wave * 16 + r15;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const bool qvalid = qi < w.nq;
  const long qrow = q_row(p, w, qvalid ? qi : 0);
  const T* qptr = qp ? qp + qrow * p.C + head * HD : qkv + qrow * 3 * p.C + head * HD;
  const T* doptr = reinterpret_cast<const T*>(p.dout) + qrow * p.C + head * HD;
  const T* optr = reinterpret_cast<const T*>(p.out) + qrow * p.C + head * HD;

  zero_images<T, HD>(smem, 2 * A::ROW_BYTES);
  typename A::Frag qf[A::KS], dof[A::KS];
#pragma unroll
  for (int s = 0; s < A::KS; ++s) {
    qf[s] = load_row_frag_global<T, HD>(qptr, s, q);
    dof[s] = load_row_frag_global<T, HD>(doptr, s, q);
  }
  // delta: each of the 4 lanes sharing a row sums a quarter of d
  float delta = 0.f;   // from 16-byte fragment loads (dO's is already in registers), not element loads
#pragma unroll
  for (int s = 0; s < A::KS; ++s) delta = frag_dot(dof[s], load_row_frag_global<T, HD>(optr, s, q), delta);
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const float lse = qvalid ? p.lse[qrow * p.heads + head] : 0.f;
  if (qvalid && q == 0) p.delta[qrow * p.heads + head] = delta;

  f32x4 dq[A::DB];
#pragma unroll
  for (int db = 0; db < A::DB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);
  const int ntiles = (nkeys + 63) >> 6;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    if (tid < 64) {
      const int c = t * 64 + tid;
      const T* kp = nullptr;
      float b = NEG_BIG;
      if (c < w.nvalid) { kp = qkv + key_row(p, w, c) * 3 * p.C + p.C + head * HD; b = 0.f; }
      else if (c == w.nvalid && w.npad > 0) { kp = reinterpret_cast<const T*>(p.bias) + p.C + head * HD; b = __logf((float)w.npad); }
      kptr[tid] = kp; kb[tid] = b;
    }
    __syncthreads();
#if SPG_DQ_STAGE_TILES
    {
      const StageSrc kv[2] = {{reinterpret_cast<const void* const*>(kptr), 0, kimg}, {reinterpret_cast<const void* const*>(kptr), p.C, vimg}};
      stage_tiles<T, HD, 2>(kv, qkv);
    }
#else
    stage_tile<T, HD>(kptr, 0, kimg);
    stage_tile<T, HD>(kptr, p.C, vimg);
#endif
    __syncthreads();
    f32x4 sacc[4], pacc[4];
    mma_scores<T, HD>(kimg, qf, lane, sacc);
    mma_scores<T, HD>(vimg, dof, lane, pacc);
    float ds[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + nb * 16 + q * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = __expf(sacc[nb][r] * p.scale + b4[r] - lse);
        if constexpr (SUB) { if (!same_sub(p, qi, t * 64 + nb * 16 + q * 4 + r)) pr = 0.f; }
        ds[nb][r] = pr * (pacc[nb][r] - delta);
      }
    }
    mma_over_tokens<T, HD>(kimg, ds, lane, dq);
  }
  if (qvalid) {
    T* dst = p.qp ? reinterpret_cast<T*>(p.dqp) + qrow * p.C + head * HD
                  : reinterpret_cast<T*>(p.dqkv) + qrow * 3 * p.C + head * HD;
    store_rows_T<T, HD>(dst, dq, p.scale, lane);
  }
}

// =====================================================================================================
// backward, key side: dK, dV (keys stationary, queries streamed)
// =====================================================================================================
template <typename T, int HD, bool SUB = false>   // SUB: packed 4 x 4 windows (block mask same_sub)
__global__ __launch_bounds__(AT) __attribute__((amdgpu_waves_per_eu(SUB ? 3 : 1, 3))) void attn_bwd_dkv_kernel(AttnP p) {   // (SUB: held to 168 VGPRs, three waves per SIMD)
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;
  char* doimg = qimg + A::ROW_BYTES;
  const T** qptrs = reinterpret_cast<const T**>(doimg + A::ROW_BYTES);
  const T** doptrs = qptrs + 64;
  float* lse_s = reinterpret_cast<float*>(doptrs + 64);
  float* delta_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const int head = blockIdx.y;
  const Win w = get_win(p, blockIdx.z);
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);
  if (blockIdx.x * 64 >= nkeys) return;
  const int c = blockIdx.x * 64 + wave * 16 + r15;  // this lane's key
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const T* kp = nullptr;
  float kbias = NEG_BIG;
  long krow = 0;
  if (c < w.nvalid) { krow = key_row(p, w, c); kp = qkv + krow * 3 * p.C + p.C + head * HD; kbias = 0.f; }
  else if (c == w.nvalid && w.npad > 0) { kp = reinterpret_cast<const T*>(p.bias) + p.C + head * HD; kbias = __logf((float)w.npad); }

  zero_images<T, HD>(smem, 2 * A::ROW_BYTES);
  typename A::Frag kf[A::KS], vf[A::KS];
#pragma unroll
  for (int s = 0; s < A::KS; ++s) {
    kf[s] = load_row_frag_global<T, HD>(kp, s, q);
    vf[s] = load_row_frag_global<T, HD>(kp ? kp + p.C : nullptr, s, q);
  }
  f32x4 dk[A::DB], dv[A::DB];
#pragma unroll
  for (int db = 0; db < A::DB; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int ntiles = (w.nq + 63) >> 6;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    if (tid < 64) {
      const int i = t * 64 + tid;
      const T* a = nullptr; const T* b = nullptr;
      float ls = 1.0e30f, dl = 0.f;
      if (i < w.nq) {
        const long row = q_row(p, w, i);
        a = qp ? qp + row * p.C + head * HD : qkv + row * 3 * p.C + head * HD;
        b = reinterpret_cast<const T*>(p.dout) + row * p.C + head * HD;
        ls = p.lse[row * p.heads + head];
        dl = p.delta[row * p.heads + head];
      }
      qptrs[tid] = a; doptrs[tid] = b; lse_s[tid] = ls; delta_s[tid] = dl;
    }
    __syncthreads();
#if SPG_DKV_STAGE_TILES
    {
      const StageSrc qd[2] = {{reinterpret_cast<const void* const*>(qptrs), 0, qimg}, {reinterpret_cast<const void* const*>(doptrs), 0, doimg}};
      stage_tiles<T, HD, 2>(qd, qkv);
    }
#else
    stage_tile<T, HD>(qptrs, 0, qimg);     // (not stage_tiles: this kernel sits at exactly 168 VGPRs = three waves per SIMD; the batched form takes 180: 52.2 -> 55.8 us)
    stage_tile<T, HD>(doptrs, 0, doimg);
#endif
    __syncthreads();
    f32x4 sacc[4], pacc[4];
    mma_scores<T, HD>(qimg, kf, lane, sacc);   // S[i][key]
    mma_scores<T, HD>(doimg, vf, lane, pacc);  // dP[i][key]
    float pr[4][4], ds[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + nb * 16 + q * 4);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + nb * 16 + q * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[nb][r] = __expf(sacc[nb][r] * p.scale + kbias - l4[r]);
        if constexpr (SUB) { if (!same_sub(p, t * 64 + nb * 16 + q * 4 + r, c)) pr[nb][r] = 0.f; }
        ds[nb][r] = pr[nb][r] * (pacc[nb][r] - d4[r]);
      }
    }
    mma_over_tokens<T, HD>(doimg, pr, lane, dv);
    mma_over_tokens<T, HD>(qimg, ds, lane, dk);
  }
  if (c < w.nvalid) {
    T* dst = reinterpret_cast<T*>(p.dqkv) + krow * 3 * p.C + p.C + head * HD;
    store_rows_T<T, HD>(dst, dk, p.scale, lane);
    store_rows_T<T, HD>(dst + p.C, dv, 1.f, lane);
  } else if (c == w.nvalid && w.npad > 0) {
    float* db_ = p.dbias + p.C + head * HD;
#pragma unroll
    for (int db = 0; db < A::DB; ++db) {
      const int d = db * 16 + q * 4;
      if (d < HD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(db_ + d + r, dk[db][r] * p.scale);
          atomicAdd(db_ + p.C + d + r, dv[db][r]);
        }
      }
    }
  }
}

// =====================================================================================================
// backward in ONE pass for windows of at most 64 keys and 64 queries (Hiera-L at 384 px: stage 1, stage 2's packed 4 x 4 windows, the two
// transition blocks in front of them, stage 4): one workgroup per (window, head) stages K, V, Q and dO ONCE as four row images, then every
// wave runs the query-side program of attn_bwd_dq_kernel for its 16 queries and the key-side program of attn_bwd_dkv_kernel for its 16
// keys, with the stationary fragments read from the images instead of global memory.  Same products in the same order as the two kernels
// (bit-identical gradients); what goes away is the second launch and the second read of qkv / dO / lse and the delta round trip through
// memory: at stage 1 the pair moved 275 MB per block for 170 MB of operands and results (42 + 66 us).
// =====================================================================================================
template <typename T, int HD, bool SUB = false>
__global__ __launch_bounds__(AT) void attn_bwd_small_kernel(AttnP p) {
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = kimg + A::ROW_BYTES;
  char* qimg = vimg + A::ROW_BYTES;
  char* doimg = qimg + A::ROW_BYTES;
  const T** kptr = reinterpret_cast<const T**>(doimg + A::ROW_BYTES);
  const T** qptrs = kptr + 64;
  const T** doptrs = qptrs + 64;
  float* kb = reinterpret_cast<float*>(doptrs + 64);
  float* lse_s = kb + 64;
  float* delta_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const int head = blockIdx.y;
  const Win w = get_win(p, blockIdx.z);
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);   // (launcher: <= 64, and nq <= 64)

  zero_images<T, HD>(smem, 4 * A::ROW_BYTES);
  if (tid < 64) {
    const int c = tid;
    const T* kp = nullptr;
    float b = NEG_BIG;
    if (c < w.nvalid) { kp = qkv + key_row(p, w, c) * 3 * p.C + p.C + head * HD; b = 0.f; }
    else if (c == w.nvalid && w.npad > 0) { kp = reinterpret_cast<const T*>(p.bias) + p.C + head * HD; b = __logf((float)w.npad); }
    kptr[tid] = kp; kb[tid] = b;
  } else if (tid < 128) {
    const int i = tid - 64;
    const T* a = nullptr; const T* b = nullptr;
    float ls = 1.0e30f;
    if (i < w.nq) {
      const long row = q_row(p, w, i);
      a = qp ? qp + row * p.C + head * HD : qkv + row * 3 * p.C + head * HD;
      b = reinterpret_cast<const T*>(p.dout) + row * p.C + head * HD;
      ls = p.lse[row * p.heads + head];
    }
    qptrs[i] = a; doptrs[i] = b; lse_s[i] = ls;
  }
  __syncthreads();
  {
    const StageSrc all4[4] = {{reinterpret_cast<const void* const*>(kptr), 0, kimg}, {reinterpret_cast<const void* const*>(kptr), p.C, vimg},
                              {reinterpret_cast<const void* const*>(qptrs), 0, qimg}, {reinterpret_cast<const void* const*>(doptrs), 0, doimg}};
    stage_tiles<T, HD, 4>(all4, qkv);
  }
  __syncthreads();

  // ---------------- query side (this lane's query: qi) ----------------
  const int qi = wave * 16 + r15;
  const bool qvalid = qi < w.nq;
  const long qrow = q_row(p, w, qvalid ? qi : 0);
  {
    typename A::Frag qf[A::KS], dof[A::KS];
#pragma unroll
    for (int s = 0; s < A::KS; ++s) {
      qf[s] = load_row_frag_lds<T, HD>(qimg, qi, s, q);
      dof[s] = load_row_frag_lds<T, HD>(doimg, qi, s, q);
    }
    const T* optr = qvalid ? reinterpret_cast<const T*>(p.out) + qrow * p.C + head * HD : nullptr;
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < A::KS; ++s) delta = frag_dot(dof[s], load_row_frag_global<T, HD>(optr, s, q), delta);
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    if (q == 0) delta_s[qi] = qvalid ? delta : 0.f;
    const float lse = qvalid ? lse_s[qi] : 0.f;
    f32x4 dq[A::DB];
#pragma unroll
    for (int db = 0; db < A::DB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sacc[4], pacc[4];
    mma_scores<T, HD>(kimg, qf, lane, sacc);
    mma_scores<T, HD>(vimg, dof, lane, pacc);
    float ds[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + nb * 16 + q * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = __expf(sacc[nb][r] * p.scale + b4[r] - lse);
        if constexpr (SUB) { if (!same_sub(p, qi, nb * 16 + q * 4 + r)) pr = 0.f; }
        ds[nb][r] = pr * (pacc[nb][r] - delta);
      }
    }
    mma_over_tokens<T, HD>(kimg, ds, lane, dq);
    if (qvalid) {
      T* dst = p.qp ? reinterpret_cast<T*>(p.dqp) + qrow * p.C + head * HD
                    : reinterpret_cast<T*>(p.dqkv) + qrow * 3 * p.C + head * HD;
      store_rows_T<T, HD>(dst, dq, p.scale, lane);
    }
  }
  __syncthreads();          // delta_s complete

  // ---------------- key side (this lane's key: c) ----------------
  if (wave * 16 < nkeys) {  // (wave-uniform: key blocks past the window's keys have nothing to do)
    const int c = wave * 16 + r15;
    const float kbias = kb[c];
    typename A::Frag kf[A::KS], vf[A::KS];
#pragma unroll
    for (int s = 0; s < A::KS; ++s) {
      kf[s] = load_row_frag_lds<T, HD>(kimg, c, s, q);
      vf[s] = load_row_frag_lds<T, HD>(vimg, c, s, q);
    }
    f32x4 dk[A::DB], dv[A::DB];
#pragma unroll
    for (int db = 0; db < A::DB; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 sacc[4], pacc[4];
    mma_scores<T, HD>(qimg, kf, lane, sacc);   // S[i][key]
    mma_scores<T, HD>(doimg, vf, lane, pacc);  // dP[i][key]
    float pr[4][4], ds[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + nb * 16 + q * 4);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + nb * 16 + q * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[nb][r] = __expf(sacc[nb][r] * p.scale + kbias - l4[r]);
        if constexpr (SUB) { if (!same_sub(p, nb * 16 + q * 4 + r, c)) pr[nb][r] = 0.f; }
        ds[nb][r] = pr[nb][r] * (pacc[nb][r] - d4[r]);
      }
    }
    mma_over_tokens<T, HD>(doimg, pr, lane, dv);
    mma_over_tokens<T, HD>(qimg, ds, lane, dk);
    if (c < w.nvalid) {
      const long krow = key_row(p, w, c);
      T* dst = reinterpret_cast<T*>(p.dqkv) + krow * 3 * p.C + p.C + head * HD;
      store_rows_T<T, HD>(dst, dk, p.scale, lane);
      store_rows_T<T, HD>(dst + p.C, dv, 1.f, lane);
    } else if (c == w.nvalid && w.npad > 0) {
      float* db_ = p.dbias + p.C + head * HD;
#pragma unroll
      for (int db = 0; db < A::DB; ++db) {
        const int d = db * 16 + q * 4;
        if (d < HD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(db_ + d + r, dk[db][r] * p.scale);
            atomicAdd(db_ + p.C + d + r, dv[db][r]);
          }
        }
      }
    }
  }
}

// =====================================================================================================
// "resident window" variants (bf16, 64 < keys <= 320, queries <= 256: the stage-3 windows and block 44 of Hiera-L).
// One workgroup of 8 waves per (window, head): the whole window's K and V (forward, dQ) or Q and dO (dK/dV) are staged ONCE
// into LDS as row images, then every wave walks its own 16-row blocks over all tiles without any further barrier.  The
// tiled kernels above re-stage each 64-token tile in every query-tile workgroup; here each token is loaded once per head.
// =====================================================================================================
constexpr int RES_ROWS = 320;     // max staged rows (5 tiles of 64)
constexpr int RES_THREADS = 512;

// copies rows (pointer table in LDS, null = zero row) into a row image; `off2 >= 0` also copies the rows at ptr + off2 into
// rimg2 (K and V share a pointer table).  Unrolled by 5 so that all of a thread's global loads are issued before the first LDS
// store (a plain runtime loop serialises LDS-pointer read -> global load -> LDS write per item).
template <typename T, int HD>
__device__ __forceinline__ void stage_rows(const T* const* ptrs, int nrows, int off, char* rimg, int off2 = -1, char* rimg2 = nullptr) {
  constexpr int VEC = AC<T, HD>::VEC, NCH = AC<T, HD>::NCH, RS = AC<T, HD>::RS;
  constexpr int U = 5;
  const int total = nrows * NCH;
  for (int base = threadIdx.x; base < total; base += RES_THREADS * U) {
    u32x4 v[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + u * RES_THREADS;
      v[u] = u32x4{0u, 0u, 0u, 0u}; v2[u] = u32x4{0u, 0u, 0u, 0u};
      if (p < total) {
        const int ch = p % NCH, t = p / NCH;
        const T* sp = ptrs[t];
        if (sp) {
          v[u] = ld16(sp + off + ch * VEC);
          if (rimg2) v2[u] = ld16(sp + off2 + ch * VEC);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + u * RES_THREADS;
      if (p < total) {
        const int ch = p % NCH, t = p / NCH;
        *reinterpret_cast<u32x4*>(rimg + t * RS + ch * 16) = v[u];
        if (rimg2) *reinterpret_cast<u32x4*>(rimg2 + t * RS + ch * 16) = v2[u];
      }
    }
  }
}
// One-pass staging: every 16-byte chunk of rows [0, nrows) of up to two row images is written exactly once -- data chunks from the
// row pointers the callables return (null = zero row), the pad chunks of a row (d >= HD) as zeros.  No pointer table in LDS, no
// separate zero fill, no barrier before the loads (these kernels are bound by exactly that dependent chain: table -> barrier -> loads
// -> LDS stores -> barrier).  srcA / srcB: token -> const T* (row start) or nullptr; rimgB == nullptr: one image.
#ifndef SPG_STAGE_U
#define SPG_STAGE_U 7
#endif
template <typename T, int HD, typename FA, typename FB>
__device__ __forceinline__ void stage_rows2(int nrows, FA srcA, char* rimgA, FB srcB, char* rimgB) {
  constexpr int VEC = AC<T, HD>::VEC, NCH = AC<T, HD>::NCH, RS = AC<T, HD>::RS;
  constexpr int RCH = RS / 16;        // 16-byte chunks per image row incl. padding
  constexpr int U = SPG_STAGE_U;      // chunks per thread and pass: 7 x 512 threads cover a 256-row window (13 chunks a row) in ONE global round trip
  const int total = nrows * RCH;
  for (int base = threadIdx.x; base < total; base += RES_THREADS * U) {
    u32x4 v[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + u * RES_THREADS;
      v[u] = u32x4{0u, 0u, 0u, 0u}; v2[u] = u32x4{0u, 0u, 0u, 0u};
      if (p < total) {
        const int t = p / RCH, ch = p - t * RCH;
        if (ch < NCH) {
          const T* a = srcA(t);
          if (a) v[u] = ld16(a + ch * VEC);
          if (rimgB) { const T* b = srcB(t); if (b) v2[u] = ld16(b + ch * VEC); }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = base + u * RES_THREADS;
      if (p < total) {
        *reinterpret_cast<u32x4*>(rimgA + p * 16) = v[u];
        if (rimgB) *reinterpret_cast<u32x4*>(rimgB + p * 16) = v2[u];
      }
    }
  }
}
template <typename T, int HD> struct ResLds {
  static constexpr int IMG = RES_ROWS * AC<T, HD>::RS;
  static constexpr int BYTES = 2 * IMG + RES_ROWS * 8 + 3 * RES_ROWS * 4;
};

#ifdef SPG_DEV_KERNELS
// in-kernel stamps of the resident forward kernel (tools/attn_stamps.py, SPG_ATTN_STAMPS=1): per workgroup and wave, cycles in
// [start (absolute) | setup + staging + barrier | its query blocks | end (absolute)]; plus the workgroup's window size
__device__ unsigned long long attn_stamps[2048 * 8 * 4];
__device__ int attn_stamp_nq[2048];
#endif
__device__ __forceinline__ unsigned long long attn_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// One 64-key tile of a resident window for a wave's 16 query rows; NB = the tile's 16-key blocks that hold keys.  A window's last tile
// often holds the virtual pad key alone (64 / 128 valid keys + 1): NB = 1 does a quarter of the score MFMAs and exponentials and half of
// the P V steps for it.
template <typename T, int HD, int NB>
__device__ __forceinline__ void res_fwd_tile(const char* kimg, const char* vimg, const float* kb, const typename AC<T, HD>::Frag (&qf)[AC<T, HD>::KS],
                                             int lane, float scale, float& m, float& l, f32x4 (&o)[AC<T, HD>::DB]) {
  const int q = lane >> 4;
  f32x4 sacc[4];
  mma_scores<T, HD, NB>(kimg, qf, lane, sacc);
  float pv[4][4];
  float mx = NEG_BIG;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + nb * 16 + q * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { pv[nb][r] = sacc[nb][r] * scale + b4[r]; mx = fmaxf(mx, pv[nb][r]); }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mn = fmaxf(m, mx);
  const float alpha = __expf(m - mn);
  m = mn;
  float ps = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) { pv[nb][r] = __expf(pv[nb][r] - mn); ps += pv[nb][r]; }
  l = l * alpha + ps;
#pragma unroll
  for (int db = 0; db < AC<T, HD>::DB; ++db) o[db] *= alpha;
  mma_over_tokens<T, HD, NB>(vimg, pv, lane, o);
}
template <typename T, int HD, int NB>
__device__ __forceinline__ void res_dq_tile(const char* kimg, const char* vimg, const float* kb, const typename AC<T, HD>::Frag (&qf)[AC<T, HD>::KS],
                                            const typename AC<T, HD>::Frag (&dof)[AC<T, HD>::KS], int lane, float scale, float lse, float delta,
                                            f32x4 (&dq)[AC<T, HD>::DB]) {
  const int q = lane >> 4;
  f32x4 sacc[4], pacc[4];
  mma_scores<T, HD, NB>(kimg, qf, lane, sacc);
  mma_scores<T, HD, NB>(vimg, dof, lane, pacc);
  float ds[4][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + nb * 16 + q * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pr = __expf(sacc[nb][r] * scale + b4[r] - lse);
      ds[nb][r] = pr * (pacc[nb][r] - delta);
    }
  }
  mma_over_tokens<T, HD, NB>(kimg, ds, lane, dq);
}

template <typename T, int HD, bool DQ, int DBG = 0>
__device__ __forceinline__ void res_q_body(const AttnP& p, const int head, const int widx, const int part, const int nparts) {   // forward (DQ=false) or dQ (DQ=true)
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = kimg + ResLds<T, HD>::IMG;
  float* kb = reinterpret_cast<float*>(vimg + ResLds<T, HD>::IMG + RES_ROWS * 8);   // (LDS map of ResLds: two images, 8 B x rows spare, 3 float arrays)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const Win w = get_win(p, widx);
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);
  const int ntiles = (nkeys + 63) >> 6, nrows = ntiles * 64;
  const int nfull = (nkeys - (ntiles - 1) * 64) <= 16 ? ntiles - 1 : ntiles;   // a last tile of <= 16 keys takes the one-block path
  // this unit's run of query blocks: part, part + nparts, ... in steps of 8 blocks (one per wave)
  const int rb0 = part * (RES_THREADS / 64);
  if (rb0 * 16 >= w.nq) return;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  if constexpr (DBG == 1) t0 = attn_now();
  // the wave's first query block: its global loads are issued ahead of the K / V staging so both round trips overlap
  const int rbf = rb0 + wave;
  typename A::Frag qf0[A::KS], dof0[DQ ? A::KS : 1], of0[DQ ? A::KS : 1];
  float lse0 = 0.f;
  {
    const int qi = rbf * 16 + r15;
    const bool qvalid = qi < w.nq;
    const long qrow = q_row(p, w, qvalid ? qi : 0);
    const T* qptr = !qvalid ? nullptr : (qp ? qp + row1c(p, qrow) + head * HD : qkv + row3c(p, qrow) + head * HD);
#pragma unroll
    for (int s = 0; s < A::KS; ++s) qf0[s] = load_row_frag_global<T, HD>(qptr, s, q);
    if constexpr (DQ) {
      const T* doptr = qvalid ? reinterpret_cast<const T*>(p.dout) + row1c(p, qrow) + head * HD : nullptr;
      const T* optr = qvalid ? reinterpret_cast<const T*>(p.out) + row1c(p, qrow) + head * HD : nullptr;
#pragma unroll
      for (int s = 0; s < A::KS; ++s) { dof0[s] = load_row_frag_global<T, HD>(doptr, s, q); of0[s] = load_row_frag_global<T, HD>(optr, s, q); }
      if (qvalid) lse0 = p.lse[qrow * p.heads + head];
    }
  }
  // per-key score bias (0 / log(n_pad) for the virtual pad key / -inf for the unused slots of the last tile), then K and V in one pass
  for (int c = tid; c < nrows; c += RES_THREADS)
    kb[c] = c < w.nvalid ? 0.f : ((c == w.nvalid && w.npad > 0) ? __logf((float)w.npad) : NEG_BIG);
  {
    auto ksrc = [&](int c) -> const T* {
      if (c < w.nvalid) return qkv + row3c(p, key_row(p, w, c)) + p.C + head * HD;
      if (c == w.nvalid && w.npad > 0) return reinterpret_cast<const T*>(p.bias) + p.C + head * HD;
      return nullptr;
    };
    auto vsrc = [&](int c) -> const T* { const T* k = ksrc(c); return k ? k + p.C : nullptr; };
    if constexpr (DBG == 1) t1 = attn_now();
    stage_rows2<T, HD>(nrows, ksrc, kimg, vsrc, vimg);
  }
  __syncthreads();
  if constexpr (DBG == 1) t2 = attn_now();

  // (Measured and dropped, round 2: a wave taking its two query blocks of a 16 x 16 window TOGETHER -- each K / V fragment read once for
  // both, two independent chains -- changed nothing (18.5 -> 19.2 us on a 2 % slower box), like the one-pass staging above and the
  // split over workgroups: neither the staging chain nor the per-wave compute chain is what bounds these launches.  Unresolved.)
  for (int rb = rb0 + wave; rb * 16 < w.nq; rb += (RES_THREADS / 64) * nparts) {
    const int qi = rb * 16 + r15;
    const bool qvalid = qi < w.nq;
    const long qrow = q_row(p, w, qvalid ? qi : 0);
    const T* qptr = qp ? qp + row1c(p, qrow) + head * HD : qkv + row3c(p, qrow) + head * HD;
    const bool first = rb == rbf;
    typename A::Frag qf[A::KS];
#pragma unroll
    for (int s = 0; s < A::KS; ++s) qf[s] = first ? qf0[s] : load_row_frag_global<T, HD>(qptr, s, q);
    if constexpr (!DQ) {
      f32x4 o[A::DB];
#pragma unroll
      for (int db = 0; db < A::DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      float m = NEG_BIG, l = 0.f;
      for (int t = 0; t < nfull; ++t)
        res_fwd_tile<T, HD, 4>(kimg + t * A::ROW_BYTES, vimg + t * A::ROW_BYTES, kb + t * 64, qf, lane, p.scale, m, l, o);
      if (nfull < ntiles)
        res_fwd_tile<T, HD, 1>(kimg + nfull * A::ROW_BYTES, vimg + nfull * A::ROW_BYTES, kb + nfull * 64, qf, lane, p.scale, m, l, o);
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      if (qvalid) {
        T* orow = reinterpret_cast<T*>(p.out) + row1c(p, qrow) + head * HD;
        store_rows_T<T, HD>(orow, o, 1.f / l, lane);
        if (q == 0) p.lse[qrow * p.heads + head] = m + __logf(l);
      }
    } else {
      const T* doptr = reinterpret_cast<const T*>(p.dout) + row1c(p, qrow) + head * HD;
      const T* optr = reinterpret_cast<const T*>(p.out) + row1c(p, qrow) + head * HD;
      typename A::Frag dof[A::KS];
#pragma unroll
      for (int s = 0; s < A::KS; ++s) dof[s] = first ? dof0[s] : load_row_frag_global<T, HD>(doptr, s, q);
      float delta = 0.f;   // from 16-byte fragment loads (dO's is already in registers), not element loads
#pragma unroll
      for (int s = 0; s < A::KS; ++s) delta = frag_dot(dof[s], first ? of0[s] : load_row_frag_global<T, HD>(optr, s, q), delta);
      delta += __shfl_xor(delta, 16, 64);
      delta += __shfl_xor(delta, 32, 64);
      const float lse = first ? lse0 : (qvalid ? p.lse[qrow * p.heads + head] : 0.f);
      if (qvalid && q == 0) p.delta[qrow * p.heads + head] = delta;
      f32x4 dq[A::DB];
#pragma unroll
      for (int db = 0; db < A::DB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < nfull; ++t)
        res_dq_tile<T, HD, 4>(kimg + t * A::ROW_BYTES, vimg + t * A::ROW_BYTES, kb + t * 64, qf, dof, lane, p.scale, lse, delta, dq);
      if (nfull < ntiles)
        res_dq_tile<T, HD, 1>(kimg + nfull * A::ROW_BYTES, vimg + nfull * A::ROW_BYTES, kb + nfull * 64, qf, dof, lane, p.scale, lse, delta, dq);
      if (qvalid) {
        T* dst = p.qp ? reinterpret_cast<T*>(p.dqp) + row1c(p, qrow) + head * HD
                      : reinterpret_cast<T*>(p.dqkv) + row3c(p, qrow) + head * HD;
        store_rows_T<T, HD>(dst, dq, p.scale, lane);
      }
    }
  }
#ifdef SPG_DEV_KERNELS
  if constexpr (DBG == 1) {
    t3 = attn_now();
    const int wg = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;   // (unit-major)
    if (lane == 0 && wg < 2048) {
      unsigned long long* o = attn_stamps + (wg * 8 + wave) * 4;
      o[0] = t0; o[1] = t2 - t0; o[2] = t3 - t2; o[3] = t3;   // (t0 / t3 absolute: dispatch skew and the launch's span, tools/attn_stamps.py)
      if (wave == 0) attn_stamp_nq[wg] = w.nq;
    }
  }
#endif
}

template <typename T, int HD, bool DQ, int DBG = 0>
__global__ __launch_bounds__(RES_THREADS) void attn_res_q_kernel(AttnP p, ResPlan pl) {
  int widx, part, nparts;
  res_unit(pl, p, blockIdx.y, widx, part, nparts);
  res_q_body<T, HD, DQ, DBG>(p, blockIdx.x, widx, part, nparts);
}

// OWN_DELTA: delta[i] = dO_i . O_i formed here (the merged backward launch: the dQ units that would have written it run beside these)
template <typename T, int HD, bool OWN_DELTA>
__device__ __forceinline__ void res_dkv_body(const AttnP& p, const int head, const int widx, const int part, const int nparts) {
  using A = AC<T, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;
  char* doimg = qimg + ResLds<T, HD>::IMG;
  float* lse_s = reinterpret_cast<float*>(doimg + ResLds<T, HD>::IMG + RES_ROWS * 8);
  float* delta_s = lse_s + RES_ROWS;
  long* qrows = nullptr; (void)qrows;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r15 = lane & 15, q = lane >> 4;
  const Win w = get_win(p, widx);
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* qp = reinterpret_cast<const T*>(p.qp);
  const int nkeys = w.nvalid + (w.npad > 0 ? 1 : 0);
  const int ntq = (w.nq + 63) >> 6, nrows = ntq * 64;
  const int kb0 = part * (RES_THREADS / 64);                 // this unit's run of key blocks (the pad key's block goes round-robin too)
  if (kb0 * 16 >= nkeys) return;
  for (int i = tid; i < nrows; i += RES_THREADS) {
    float ls = 1.0e30f, dl = 0.f;
    if (i < w.nq) {
      const long row = q_row(p, w, i);
      ls = p.lse[row * p.heads + head];
      if constexpr (OWN_DELTA) {
        const T* dop = reinterpret_cast<const T*>(p.dout) + row1c(p, row) + head * HD;
        const T* op = reinterpret_cast<const T*>(p.out) + row1c(p, row) + head * HD;
        typename A::Frag a_[A::NCH], b_[A::NCH];
#pragma unroll
        for (int ch = 0; ch < A::NCH; ++ch) {
          if constexpr (sizeof(T) == 2) { a_[ch] = __builtin_bit_cast(bf16x8_t, ld16(dop + ch * 8)); b_[ch] = __builtin_bit_cast(bf16x8_t, ld16(op + ch * 8)); }
          else { a_[ch] = 0.f; b_[ch] = 0.f; }   // (fp32: summed element-wise below)
        }
#pragma unroll
        for (int ch = 0; ch < A::NCH; ++ch) dl = frag_dot(a_[ch], b_[ch], dl);
        if constexpr (sizeof(T) != 2) { for (int d = 0; d < HD; ++d) dl += dop[d] * op[d]; }
      } else {
        dl = p.delta[row * p.heads + head];
      }
    }
    lse_s[i] = ls; delta_s[i] = dl;
  }
  {
    auto qsrc = [&](int i) -> const T* {
      if (i >= w.nq) return nullptr;
      const long row = q_row(p, w, i);
      return qp ? qp + row1c(p, row) + head * HD : qkv + row3c(p, row) + head * HD;
    };
    auto dosrc = [&](int i) -> const T* {
      if (i >= w.nq) return nullptr;
      return reinterpret_cast<const T*>(p.dout) + row1c(p, q_row(p, w, i)) + head * HD;
    };
    stage_rows2<T, HD>(nrows, qsrc, qimg, dosrc, doimg);
  }
  __syncthreads();

  // (Measured and dropped, round 2: dealing the pad key's query tiles over the waves instead of giving one wave a second block shortens
  // the longest wave but doubles the float atomics that land on the same 144 bias-gradient addresses from every window of a head at the
  // kernel's end -- 23.3 -> 26.9 us; with the atomics compiled out 21.1.  Those same-address atomics are ~2 us of this kernel as it is.)
  for (int kbk = kb0 + wave; kbk * 16 < nkeys; kbk += (RES_THREADS / 64) * nparts) {
    const int c = kbk * 16 + r15;
    const T* kp = nullptr;
    float kbias = NEG_BIG;
    long krow = 0;
    if (c < w.nvalid) { krow = key_row(p, w, c); kp = qkv + row3c(p, krow) + p.C + head * HD; kbias = 0.f; }
    else if (c == w.nvalid && w.npad > 0) { kp = reinterpret_cast<const T*>(p.bias) + p.C + head * HD; kbias = __logf((float)w.npad); }
    typename A::Frag kf[A::KS], vf[A::KS];
#pragma unroll
    for (int s = 0; s < A::KS; ++s) {
      kf[s] = load_row_frag_global<T, HD>(kp, s, q);
      vf[s] = load_row_frag_global<T, HD>(kp ? kp + p.C : nullptr, s, q);
    }
    f32x4 dk[A::DB], dv[A::DB];
#pragma unroll
    for (int db = 0; db < A::DB; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = 0; t < ntq; ++t) {
      f32x4 sacc[4], pacc[4];
      mma_scores<T, HD>(qimg + t * A::ROW_BYTES, kf, lane, sacc);
      mma_scores<T, HD>(doimg + t * A::ROW_BYTES, vf, lane, pacc);
      float pr[4][4], ds[4][4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + t * 64 + nb * 16 + q * 4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + t * 64 + nb * 16 + q * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[nb][r] = __expf(sacc[nb][r] * p.scale + kbias - l4[r]);
          ds[nb][r] = pr[nb][r] * (pacc[nb][r] - d4[r]);
        }
      }
      mma_over_tokens<T, HD>(doimg + t * A::ROW_BYTES, pr, lane, dv);
      mma_over_tokens<T, HD>(qimg + t * A::ROW_BYTES, ds, lane, dk);
    }
    if (c < w.nvalid) {
      T* dst = reinterpret_cast<T*>(p.dqkv) + row3c(p, krow) + p.C + head * HD;
      store_rows_T<T, HD>(dst, dk, p.scale, lane);
      store_rows_T<T, HD>(dst + p.C, dv, 1.f, lane);
    } else if (c == w.nvalid && w.npad > 0) {
      float* db_ = p.dbias + p.C + head * HD;
#pragma unroll
      for (int db = 0; db < A::DB; ++db) {
        const int d = db * 16 + q * 4;
        if (d < HD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(db_ + d + r, dk[db][r] * p.scale);
            atomicAdd(db_ + p.C + d + r, dv[db][r]);
          }
        }
      }
    }
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(RES_THREADS) void attn_res_dkv_kernel(AttnP p, ResPlan pl) {
  int widx, part, nparts;
  res_unit(pl, p, blockIdx.y, widx, part, nparts);
  res_dkv_body<T, HD, false>(p, blockIdx.x, widx, part, nparts);
}
// dQ and dK/dV units of a resident-window backward in ONE launch, longest first across both kinds (class by class: the dQ units of
// a class, then its dK/dV units): ~2.5 rounds of workgroups packed together instead of 1.25 + 1.25 with two ragged tails, one launch less.
template <typename T, int HD>
__global__ __launch_bounds__(RES_THREADS) void attn_res_bwd_kernel(AttnP p, ResPlan plq, ResPlan plk) {
  int u = blockIdx.y, c = 0;
  bool isk = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nq_ = (i < 3 ? plq.first[i + 1] : plq.total) - plq.first[i];
    const int nk_ = (i < 3 ? plk.first[i + 1] : plk.total) - plk.first[i];
    if (u >= 0 && c >= 0) {
      if (u < nq_) { c = -1 - i; isk = false; }
      else if (u < nq_ + nk_) { u -= nq_; c = -1 - i; isk = true; }
      else u -= nq_ + nk_;
    }
  }
  c = -1 - c;                                     // the unit's class; u: its index inside (class, kind)
  int widx, part, nparts;
  if (!isk) { res_unit(plq, p, plq.first[c] + u, widx, part, nparts); res_q_body<T, HD, true, 0>(p, blockIdx.x, widx, part, nparts); }
  else { res_unit(plk, p, plk.first[c] + u, widx, part, nparts); res_dkv_body<T, HD, true>(p, blockIdx.x, widx, part, nparts); }
}

#ifndef SPG_ATTN_SMALL_BWD   // (A/B builds: -DSPG_ATTN_SMALL_BWD=0 keeps the two-kernel backward for the small windows)
#define SPG_ATTN_SMALL_BWD 1
#endif
static inline bool small_bwd_enabled() { return SPG_ATTN_SMALL_BWD != 0; }

// ---------------------------------------------------------------------------------------------------
template <typename T, int HD>
static int launch_attn(int which, AttnP p, int maxq, int maxk, hipStream_t s) {
  using A = AC<T, HD>;
  const int nwin = p.B * p.nwy * p.nwx;
#ifdef SPG_DEV_KERNELS   // A/B runs of tools/attn_bench.py only: the product library reads no environment variable
  static int use_res = -1;
  if (use_res < 0) { const char* e = getenv("SPG_ATTN"); use_res = (e && strcmp(e, "tiled") == 0) ? 0 : 1; }
#else
  constexpr int use_res = 1;
#endif
  if constexpr (sizeof(T) == 2 && ResLds<T, HD>::BYTES <= 160 * 1024) {
    const long nrows_ = (long)p.B * p.H * p.W;   // (the resident kernels index rows with 24-bit multiplies and 32-bit element offsets)
    if (use_res && maxk > 65 && maxk <= RES_ROWS && maxq <= 256 && nrows_ < (1L << 24) && nrows_ * 3 * p.C < (1L << 32)) {   // multi-tile windows only (stage 3, block 44)
      constexpr int LDS = ResLds<T, HD>::BYTES;
      static bool attr = false;
      if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res_q_kernel<T, HD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res_q_kernel<T, HD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res_dkv_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res_bwd_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
      }
      // units in longest-first order (ResPlan): a full 16 x 16 window is two workgroups.  (Round 2, first attempt: the same split as a
      // grid z dimension was SLOWER, 58 -> 66 us per backward pair -- z-major dispatch put the second halves of the big windows LAST,
      // behind 192 empty workgroups, on CUs that had already run a small window.)
      const ResPlan plq = res_plan(p, false), plk = res_plan(p, true);
      const dim3 grid(p.heads, plq.total, 1), gridk(p.heads, plk.total, 1);
      if (which == 0) {
#ifdef SPG_DEV_KERNELS
        static const char* st_ = getenv("SPG_ATTN_STAMPS");
        if (st_ && atoi(st_) == 1) {
          hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res_q_kernel<T, HD, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          hipLaunchKernelGGL((attn_res_q_kernel<T, HD, false, 1>), grid, dim3(RES_THREADS), LDS, s, p, plq);
          return check_launch("attn_fwd(res, stamps)");
        }
#endif
        hipLaunchKernelGGL((attn_res_q_kernel<T, HD, false>), grid, dim3(RES_THREADS), LDS, s, p, plq);
        return check_launch("attn_fwd(res)");
      }
#ifdef SPG_DEV_KERNELS
      static const char* sp_ = getenv("SPG_ATTN_BWD_SPLIT");   // A/B: the two-launch backward
      if (sp_ && atoi(sp_) == 1) {
        hipLaunchKernelGGL((attn_res_q_kernel<T, HD, true>), grid, dim3(RES_THREADS), LDS, s, p, plq);
        int rc = check_launch("attn_bwd_dq(res)");
        if (rc) return rc;
        hipLaunchKernelGGL((attn_res_dkv_kernel<T, HD>), gridk, dim3(RES_THREADS), LDS, s, p, plk);
        return check_launch("attn_bwd_dkv(res)");
      }
#endif
      hipLaunchKernelGGL((attn_res_bwd_kernel<T, HD>), dim3(p.heads, plq.total + plk.total, 1), dim3(RES_THREADS), LDS, s, p, plq, plk);
      return check_launch("attn_bwd(res)");
    }
  }
  const bool sub = p.sub > 0;
  if (which == 0) {
    const size_t lds = 2 * A::ROW_BYTES + 64 * 8 + 64 * 4;
    if (sub) hipLaunchKernelGGL((attn_fwd_kernel<T, HD, true>), dim3(cdiv(maxq, 64), p.heads, nwin), dim3(AT), lds, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<T, HD, false>), dim3(cdiv(maxq, 64), p.heads, nwin), dim3(AT), lds, s, p);
    return check_launch("attn_fwd");
  }
  // every window at most 64 keys (incl. the virtual pad key: a window of ws x wsx = 64 slots is either full or has fewer valid keys + 1)
  // and 64 queries: one pass
  if (small_bwd_enabled() && maxq <= 64 && (sub ? maxk - 1 : (p.ws * p.wsx <= 64 ? p.ws * p.wsx : maxk)) <= 64) {
    const size_t lds = 4 * A::ROW_BYTES + 3 * 64 * 8 + 3 * 64 * 4;
    static bool attr_small = false;
    if (lds > 65536 && !attr_small) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small_kernel<T, HD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small_kernel<T, HD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_small = true;
    }
    if (sub) hipLaunchKernelGGL((attn_bwd_small_kernel<T, HD, true>), dim3(1, p.heads, nwin), dim3(AT), lds, s, p);
    else hipLaunchKernelGGL((attn_bwd_small_kernel<T, HD, false>), dim3(1, p.heads, nwin), dim3(AT), lds, s, p);
    return check_launch("attn_bwd(small)");
  }
  {
    const size_t lds = 2 * A::ROW_BYTES + 64 * 8 + 64 * 4;
    if (sub) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, true>), dim3(cdiv(maxq, 64), p.heads, nwin), dim3(AT), lds, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD, false>), dim3(cdiv(maxq, 64), p.heads, nwin), dim3(AT), lds, s, p);
    int rc = check_launch("attn_bwd_dq");
    if (rc) return rc;
  }
  {
    const size_t lds = 2 * A::ROW_BYTES + 2 * 64 * 8 + 2 * 64 * 4;
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<T, HD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<T, HD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    // (packed windows have no padded slots: exactly maxk - 1 keys, one key tile)
    const int kx = cdiv(sub ? maxk - 1 : maxk, 64);
    if (sub) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, true>), dim3(kx, p.heads, nwin), dim3(AT), lds, s, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD, false>), dim3(kx, p.heads, nwin), dim3(AT), lds, s, p);
    return check_launch("attn_bwd_dkv");
  }
}

static int fill_geom(AttnP& p, int B, int H, int W, int heads, int hd, int ws, bool pooled, int* maxq, int* maxk) {
  p.B = B; p.H = H; p.W = W; p.heads = heads; p.C = heads * hd;
  if (ws <= 0) ws = H > W ? H : W;
  if (pooled && ((H | W | ws) & 1)) { set_error("attn: pooled queries need even H, W, window (H=%d W=%d ws=%d)", H, W, ws); return SPG_ERR_BAD_ARG; }
  // 4 x 4 windows that tile the map exactly (stage 2 of Hiera at any multiple-of-64 image size): four windows side by side are
  // processed as ONE 4 x 16 'window' with a block mask (same_sub) -- a full 64-token tile and four busy waves per workgroup
  const bool pack = ws == 4 && H % 4 == 0 && W % 16 == 0;
  const int wsx = pack ? 16 : ws;
  p.ws = ws; p.wsx = wsx; p.sub = pack ? 4 : 0;
  p.nwy = cdiv(H, ws); p.nwx = cdiv(W, wsx);
  if (pooled) { p.Hq = H / 2; p.Wq = W / 2; p.wsq = ws / 2; p.wsxq = wsx / 2; }
  else { p.Hq = H; p.Wq = W; p.wsq = ws; p.wsxq = wsx; }
  p.scale = 1.0f / sqrtf((float)hd);
  const int hv = ws < H ? ws : H, wv = wsx < W ? wsx : W;
  *maxk = hv * wv + 1;
  *maxq = pooled ? (hv / 2) * (wv / 2) : hv * wv;
  return SPG_OK;
}

template <typename T>
static int dispatch_hd(int which, AttnP p, int hd, int maxq, int maxk, hipStream_t s) {
  switch (hd) {
    case 72: return launch_attn<T, 72>(which, p, maxq, maxk, s);
    case 16: return launch_attn<T, 16>(which, p, maxq, maxk, s);
    case 32: return launch_attn<T, 32>(which, p, maxq, maxk, s);
    default: set_error("attn: unsupported head_dim %d (built: 16, 32, 72)", hd); return SPG_ERR_UNSUPPORTED;
  }
}

}  // namespace spg

using namespace spg;

#ifdef SPG_DEV_KERNELS
extern "C" int spg_dev_attn_stamps(unsigned long long* out, int* nq) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::attn_stamps), sizeof(unsigned long long) * 2048 * 8 * 4) != hipSuccess) return -1;
  return hipMemcpyFromSymbol(nq, HIP_SYMBOL(spg::attn_stamp_nq), sizeof(int) * 2048) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int spg_attn_fwd(int dtype, const void* qkv, const void* q_pooled, const void* qkv_bias_t, void* out,
                            float* lse, int B, int H, int W, int heads, int hd, int ws, spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_F32 || dtype == SPG_BF16, "attn_fwd: bad dtype");
  SPG_REQUIRE(B > 0 && H > 0 && W > 0 && heads > 0, "attn_fwd: empty problem");
  AttnP p{};
  p.qkv = qkv; p.qp = q_pooled; p.bias = qkv_bias_t; p.out = out; p.lse = lse;
  int maxq, maxk;
  int rc = fill_geom(p, B, H, W, heads, hd, ws, q_pooled != nullptr, &maxq, &maxk);
  if (rc) return rc;
  return dtype == SPG_BF16 ? dispatch_hd<bf16_t>(0, p, hd, maxq, maxk, (hipStream_t)stream)
                           : dispatch_hd<float>(0, p, hd, maxq, maxk, (hipStream_t)stream);
}

extern "C" int spg_attn_bwd(int dtype, const void* qkv, const void* q_pooled, const void* qkv_bias_t, const void* out,
                            const void* dout, const float* lse, void* dqkv, void* dq_pooled, float* dbias_pad,
                            float* delta_ws, int B, int H, int W, int heads, int hd, int ws, spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_F32 || dtype == SPG_BF16, "attn_bwd: bad dtype");
  SPG_REQUIRE(B > 0 && H > 0 && W > 0 && heads > 0, "attn_bwd: empty problem");
  SPG_REQUIRE((q_pooled == nullptr) == (dq_pooled == nullptr), "attn_bwd: q_pooled and dq_pooled must both be given or both be null");
  AttnP p{};
  p.qkv = qkv; p.qp = q_pooled; p.bias = qkv_bias_t; p.out = const_cast<void*>(out); p.lse = const_cast<float*>(lse);
  p.dout = dout; p.dqkv = dqkv; p.dqp = dq_pooled; p.dbias = dbias_pad; p.delta = delta_ws;
  int maxq, maxk;
  int rc = fill_geom(p, B, H, W, heads, hd, ws, q_pooled != nullptr, &maxq, &maxk);
  if (rc) return rc;
  return dtype == SPG_BF16 ? dispatch_hd<bf16_t>(1, p, hd, maxq, maxk, (hipStream_t)stream)
                           : dispatch_hd<float>(1, p, hd, maxq, maxk, (hipStream_t)stream);
}
