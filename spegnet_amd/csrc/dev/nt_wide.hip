// Dense bf16  C[M][N] = X[M][K] W[N][K]^T (+ bias, GELU family epilogues)  for the WIDE outputs of the trunk (qkv, fc1, the fc2 dgrad:
// N a multiple of 192 with a short K loop): tiles of 192 columns and a per-problem HEIGHT of 128..224 rows, one per CU, on the
// two-wave-group schedule of conv_halo.hip.  Replaces nn.Linear of the sam2 Hiera MultiScaleBlock / MLP (reference
// models/feature_encoding.py:156-159 -> sam2 hieradet) on the bf16 path for these shapes; every other shape stays on
// gemm_nt_v3 / gemm_nt_pipe (gemm.hip).
//
// Why: at batch 8 these launches are bound by how many bytes a CU must pull through its L2 -> LDS path for its share of the output
// (~55-70 KB/us per CU) and by tile quantisation: M = 4608 x N = 2304 in 128 x 128 tiles is 648 tiles on 512 slots.  A (hm x 16) x 192
// tile needs (1/(16 hm) + 1/192) bytes per FLOP instead of (1/128 + 1/128), and the row blocks are cut so that row blocks x column tiles
// just fills the CUs (N = 2304: 12 column tiles x 21 row blocks of 13-14 sixteen-row fragments = 252 tiles on 256 CUs).
// The tile is bounded by LDS: three K = 64 steps of (14 + 12) fragments x 2 KiB = 156 KiB.
//
// Geometry
//   * workgroup = 8 waves = two groups of four (waves 0-3 / 4-7, one wave of each group per SIMD) running the same program one barrier
//     apart (cdna_hip_programming.md "The 256^2 8-phase template"): while one group issues its MFMA cluster the other issues fragment
//     reads and LDS-DMA pieces.
//   * group g owns the first ceil(hm / 2) (g = 0) or the remaining (g = 1) m-fragments of the tile, wave wq = wave & 3 of a group the
//     columns 48 wq .. +47 (3 n-fragments): up to 7 x 3 accumulator fragments per wave, 21 MFMAs per phase.
//   * a STEP is K = 64 (LDS rows of 128 B = whole cache lines per row, chunk c of row r stored at c ^ (r & 7), as in conv_halo.hip), a
//     PHASE is one K = 32 half of it.  Ring of 3 step slots, the fill runs 2 steps ahead: the even phase of step s issues the X pieces
//     of step s + 2 (4 per wave of group 0, 3 of group 1: 28 pieces), the odd phase its W pieces (3 per wave) and then waits (vmcnt(7 | 6): exactly those stay in flight)
//     for this wave's pieces of step s + 1.
//   * W rows are permuted on their way into LDS: row r of n-fragment f of wave column wq holds W[col0 + 48 wq + 12 (r >> 2) + 4 f + (r & 3)],
//     so a lane's three accumulator fragments hold 12 consecutive output columns of one row: a 16-byte and an 8-byte store, 96 contiguous
//     bytes per row and wave.
//   * MFMA roles: A operand = W (n), B operand = X (m)  =>  D[n][m]: a lane holds 4 consecutive n of one output row.
//
// RESULT (round 3, one MI355X; tools/ntw_check.py, tools/ntw_stamps.py): correct on every shape it accepts, NOT faster where it counts.
// Isolated (hipGraph of 10 launches): fc1 + GELU' 27.1 us against 28.0 for gemm_nt_v3, fc2-dgrad x gelu_h 23.9 / 25.4, plain 2304-wide
// 21.1 / 23.3, qkv (N = 1728) 18.8 / 15.9; whole step 23.45-23.49 ms with it, 23.42-23.61 without (two A/B pairs on one box).
// Stamps of M 4608 x N 2304 x K 576: a phase (21 MFMAs per wave) takes 1270 cycles -- cluster 546 (26 cycles per MFMA, the same pace as
// conv_halo.hip: with fragment reads and DMA issue of the partner wave on the same SIMD the matrix pipe is 60 % fed), issue segment 363,
// barriers 360 -- so the 18 phases are 22.8 k of the workgroup's 35.7 k cycles: 36 % of a 9-step kernel is the cold 104 KiB prologue
// burst (~9 k cycles with every CU's workgroup in it at once, MI355X_MICROARCH.md), the epilogue and the drain, and with ONE tile per CU
// nothing overlaps them.  gemm_nt_v3's two small workgroups per CU hide exactly that under each other's MFMAs.  Dev builds only
// (SPG_NT_WIDE=1); kept as the measured answer to "would a 256-row / stream-K tile fix the M = 4608 GEMMs".
//
// Synchronisation: as conv_halo.hip.  Slot (s + 2) % 3 was last read in the odd phase of step s - 1 (every wave's reads returned before
// the barrier that ended that phase's issue segment).  Loads the compiler knows of (bias, gelu_h) need no extra care: its own waits count
// only its loads, so with the inline-asm DMA operations in the queue they are stricter than necessary, never weaker.
#ifdef SPG_DEV_KERNELS
#include <algorithm>
#include <type_traits>
#include "../common.h"

namespace spg {

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned wbufvec_t;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned wbufvec2_t;
typedef __attribute__((ext_vector_type(4))) unsigned wrsrc_words_t;

constexpr int NTW_MF = 14;                              // m-fragments (16 rows) per tile at most
constexpr int NTW_NF = 12;                              // n-fragments per tile: 192 columns
constexpr int NTW_BN = NTW_NF * 16;
constexpr int NTW_A_BYTES = NTW_MF * 2048;
constexpr int NTW_B_BYTES = NTW_NF * 2048;
constexpr int NTW_SLOT = NTW_A_BYTES + NTW_B_BYTES;     // one K = 64 step: 52 KiB
constexpr int NTW_RING = 3 * NTW_SLOT;                  // 156 KiB
constexpr unsigned NTW_DEAD = 0x80000000u;              // source offset of a piece that must fill zeros (>= every operand size)
enum { NTW_ACT_NONE = 0, NTW_ACT_GELU = 1, NTW_ACT_GELU_D = 3, NTW_ACT_MULH = 4 };   // (the PIPE_ACT_* codes of gemm.hip)

#ifdef SPG_DEV_KERNELS
__device__ unsigned long long ntw_stamps[256 * 8 * 6];   // DBG 5: per workgroup and wave: cycles in [issue | barrier 1 | MFMA | barrier 2], phases, whole kernel
#endif

struct NtwArgs {
  const bf16_t* X; const bf16_t* W; bf16_t* C; bf16_t* C2; const bf16_t* Hh; const float* bias;
  int M, N, K, ldx, ldc;
  int R, T, mfrags, ntiles;      // row blocks, column tiles (N / 192), ceil(M / 16), R x T
  unsigned xbytes, wbytes, cbytes;
};

__device__ __forceinline__ wrsrc_words_t ntw_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  wrsrc_words_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
// LDS-DMA piece: 64 lanes x 16 B -> LDS[lds_addr .. +1024), lane l at +16 l; source = base + voff (per lane, range-checked: out of range
// fills zeros) + soff (wave-uniform K offset).  Inline asm: see conv_halo.hip.
__device__ __forceinline__ void ntw_dma16(wrsrc_words_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// Register loads the compiler must not track (its waits for a tracked load count only tracked loads: with the DMA operations in the queue
// they come out as vmcnt(0..2), draining the fill pipeline): the value is valid only behind one of this file's own counted waits, and
// the destination is pinned there (NTW_PIN_V) so that no use is scheduled above the wait.
__device__ __forceinline__ void ntw_load16(u32x4& dst, wrsrc_words_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void ntw_load8(u32x2& dst, wrsrc_words_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
template <int N_> __device__ __forceinline__ void ntw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// DBG (dev builds): 5 = in-kernel stamps (tools/ntw_stamps.py)
template <int ACT, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_nt_wide_kernel(NtwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int r15 = lane & 15, q = lane >> 4;
  const int M = a.M, K = a.K, T = a.T;
  const int KS = K >> 6;
  const int G = (int)gridDim.x;
  unsigned long long t_begin = 0;
  if constexpr (DBG == 5) t_begin = __builtin_amdgcn_s_memtime();
  int first;
  {   // XCD-aware bijective remap: workgroups of one XCD (blockIdx % 8) take neighbouring tiles = the same row block's X rows
    const int bid = blockIdx.x;
    const int qq = G >> 3, rr = G & 7, xcd = bid & 7, i = bid >> 3;
    first = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + i;
  }
  if (first >= a.ntiles) return;
  const int my_tiles = (a.ntiles - first + G - 1) / G;
  const wrsrc_words_t xr = ntw_rsrc_words(a.X, a.xbytes), wr = ntw_rsrc_words(a.W, a.wbytes);
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, a.cbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t c2r = __builtin_amdgcn_make_buffer_rsrc(a.C2, 0, a.C2 ? a.cbytes : 0u, 0x00020000);
  const wrsrc_words_t hr = ntw_rsrc_words(a.Hh, a.Hh ? a.cbytes : 0u);
  const wrsrc_words_t br = ntw_rsrc_words(a.bias, a.bias ? (unsigned)a.N * 4u : 0u);
  const unsigned smem_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- tile ordinal -> (first m-fragment, m-fragments, those of group 0, first column); wave-uniform, divisions only at tile switches
  auto decode = [&](int ord, int& ms, int& hm, int& h0, int& col0) __attribute__((always_inline)) {
    const int t = first + ord * G;
    const int rb = t / T, ct = t - rb * T;
    ms = (rb * a.mfrags) / a.R;
    hm = ((rb + 1) * a.mfrags) / a.R - ms;
    h0 = (hm + 1) >> 1;
    col0 = ct * NTW_BN;
  };

  // ---- per-lane constants
  const unsigned frag_lane = (unsigned)(r15 * 128 + ((q ^ (r15 & 7)) << 4));        // fragment read: row r15, chunk q (K half 0) of a 16-row image
  const unsigned b_lane = frag_lane + (unsigned)(NTW_A_BYTES + wq * 3 * 2048);
  const int pr8 = lane >> 3;                                                        // a piece's row (0..7) of this lane
  const unsigned pc8 = (unsigned)(((lane & 7) ^ pr8) << 4);                         // byte offset of the source chunk this lane fetches

  // ---- stream states (wave-uniform)
  int c_ks = 0, c_t = 0;                 // compute stream: step inside its tile, tile ordinal
  int c_ms, c_hm, c_h0, c_col0;          // its tile
  int nmw, nmw_nx;                       // this wave's m-fragments in the current / next tile
  bool tile_start = false, ts_next = false;
  unsigned a_cur;                        // X fragment lane offset inside a slot for the current tile
  int f_ks = 0, f_t = 0;                 // fill stream (2 steps ahead)
  unsigned f_soff = 0;
  unsigned r_off = 0, f_off = (unsigned)(2 * NTW_SLOT);
  unsigned voffA[4], voffB[3];
  unsigned pa0, pa1, pb0, pb1;
  // epilogue parameters of the tile being multiplied
  unsigned e_base = 0;
  int e_rows = 0, e_nmw = 4;
  u32x4 eb[3];
  u32x4 eh0[ACT == NTW_ACT_MULH ? 7 : 1];
  u32x2 eh1[ACT == NTW_ACT_MULH ? 7 : 1];

  auto aim_fill = [&](bool live, int ms, int hm, int col0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ap = wave + 8 * u;                       // X piece: tile rows 8 ap .. + 7
      const int row = ms * 16 + 8 * ap + pr8;
      voffA[u] = (live && ap < 2 * hm && row < M) ? (unsigned)(row * a.ldx) * 2u + pc8 : NTW_DEAD;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int bp = wave + 8 * u;                       // W piece: image rows 8 bp .. + 7 = half of n-fragment bp / 2
      const int fa = bp >> 1, wq_ = fa / 3, f = fa - 3 * wq_;
      const int r = 8 * (bp & 1) + pr8;
      const int col = col0 + 48 * wq_ + 12 * (r >> 2) + 4 * f + (r & 3);
      voffB[u] = live ? (unsigned)(col * K) * 2u + pc8 : NTW_DEAD;
    }
  };
  auto e_load = [&]() __attribute__((always_inline)) {   // epilogue addresses + bias of the compute stream's tile
    const int mb = grp ? c_h0 : 0;
    e_nmw = grp ? c_hm - c_h0 : c_h0;
    const int row = (c_ms + mb) * 16 + r15;
    const int col = c_col0 + 48 * wq + 12 * q;
    e_rows = M - row;                                    // m-fragment i is stored iff 16 i < e_rows
    e_base = ((unsigned)row * (unsigned)a.ldc + (unsigned)col) * 2u;
#pragma unroll
    for (int f = 0; f < 3; ++f) ntw_load16(eb[f], br, (unsigned)(col + 4 * f) * 4u);
  };
  auto h_request = [&]() __attribute__((always_inline)) {   // MULH: the tile's gelu_h values, ahead of its epilogue
    if constexpr (ACT == NTW_ACT_MULH) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const bool ok = i < e_nmw && 16 * i < e_rows;
        const unsigned o = ok ? e_base + (unsigned)(i * 16 * a.ldc) * 2u : 0xFFFFFFF0u;
        ntw_load16(eh0[i], hr, o);
        ntw_load8(eh1[i], hr, ok ? o + 16u : 0xFFFFFFF0u);
      }
    }
  };

  f32x4 acc[7][3];
  bf16x8_t Af[7], Bf[3];
  auto acc_zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int f = 0; f < 3; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // ---- epilogue straight from the accumulators: 12 consecutive columns per lane and m-fragment
  auto epilogue = [&]() __attribute__((always_inline)) {   // (behind a counted wait that covers the bias / gelu_h loads)
#pragma unroll
    for (int f = 0; f < 3; ++f) asm volatile("" : "+v"(eb[f]));
    if constexpr (ACT == NTW_ACT_MULH) {
#pragma unroll
      for (int i = 0; i < 7; ++i) { asm volatile("" : "+v"(eh0[i])); asm volatile("" : "+v"(eh1[i])); }
    }
    float bv[12];
    unpack16<float>(eb[0], bv);
    unpack16<float>(eb[1], bv + 4);
    unpack16<float>(eb[2], bv + 8);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (i >= 4 && i >= e_nmw) break;
      const bool ok = i < e_nmw && 16 * i < e_rows;
      float ev[12], dv[12];
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[4 * f + e] = acc[i][f][e] + bv[4 * f + e];
      if constexpr (ACT == NTW_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 12; ++e) { dv[e] = ev[e]; ev[e] = gelu_f(ev[e]); }
      }
      if constexpr (ACT == NTW_ACT_GELU_D) {
#pragma unroll
        for (int e = 0; e < 12; ++e) gelu_both_f(ev[e], ev[e], dv[e]);
      }
      if constexpr (ACT == NTW_ACT_MULH) {
        float h[12];
        unpack16<bf16_t>(eh0[i], h);
        const u32x4 t4 = u32x4{eh1[i].x, eh1[i].y, 0u, 0u};
        float h2[8];
        unpack16<bf16_t>(t4, h2);
#pragma unroll
        for (int e = 0; e < 8; ++e) ev[e] *= h[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[8 + e] *= h2[e];
      }
      const unsigned o = ok ? e_base + (unsigned)(i * 16 * a.ldc) * 2u : 0xFFFFFFF0u;
      const u32x4 pk = pack16<bf16_t>(ev);
      const unsigned p8 = pack2bf(ev[8], ev[9]), p9 = pack2bf(ev[10], ev[11]);
      __builtin_amdgcn_raw_buffer_store_b128(wbufvec_t{pk.x, pk.y, pk.z, pk.w}, cr, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(wbufvec2_t{p8, p9}, cr, ok ? o + 16u : 0xFFFFFFF0u, 0, 0);
      if constexpr (ACT == NTW_ACT_GELU || ACT == NTW_ACT_GELU_D) {
        const u32x4 p2 = pack16<bf16_t>(dv);
        const unsigned d8 = pack2bf(dv[8], dv[9]), d9 = pack2bf(dv[10], dv[11]);
        __builtin_amdgcn_raw_buffer_store_b128(wbufvec_t{p2.x, p2.y, p2.z, p2.w}, c2r, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(wbufvec2_t{d8, d9}, c2r, ok ? o + 16u : 0xFFFFFFF0u, 0, 0);
      }
    }
  };

  // ---- prologue: steps 0 and 1 of the first tile
  decode(0, c_ms, c_hm, c_h0, c_col0);
  aim_fill(true, c_ms, c_hm, c_col0);
  nmw = grp ? c_hm - c_h0 : c_h0;
  nmw_nx = nmw;
  a_cur = frag_lane + (unsigned)((grp ? c_h0 : 0) * 2048);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const unsigned d = smem_base + (unsigned)(s * NTW_SLOT + wave * 1024);
#pragma unroll
    for (int u = 0; u < 3; ++u) ntw_dma16(xr, d + (unsigned)(u * 8192), voffA[u], (unsigned)(s * 128));
    if (grp == 0) ntw_dma16(xr, d + (unsigned)(3 * 8192), voffA[3], (unsigned)(s * 128));   // (pieces 28..31 do not exist: the X image ends at piece 27)
#pragma unroll
    for (int u = 0; u < 3; ++u) ntw_dma16(wr, d + (unsigned)(NTW_A_BYTES + u * 8192), voffB[u], (unsigned)(s * 128));
  }
  f_ks = 2;
  f_soff = 256u;
  e_load();
  pa0 = a_cur; pa1 = pa0 ^ 64u;
  pb0 = b_lane; pb1 = pb0 ^ 64u;
  acc_zero();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  unsigned long long st_sum0 = 0, st_sum1 = 0, st_sum2 = 0, st_sum3 = 0, st_n = 0;
#define NTW_PIN_V(x) asm volatile("" : "+v"(x))
#define NTW_PIN_S(x) asm volatile("" : "+s"(x))
  // ---- one phase = K half ODD of the compute stream's step.  What later issue segments need is formed in the gaps of the MFMA
  // clusters, one to three instructions per gap, pinned there (conv_halo.hip explains why).
  auto phase = [&](auto ODD_) __attribute__((always_inline)) {
    constexpr bool ODD = decltype(ODD_)::value;
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
    if constexpr (DBG == 5) tA = __builtin_amdgcn_s_memtime();
    // ================= issue segment
#pragma unroll
    for (int f = 0; f < 3; ++f) Bf[f] = *reinterpret_cast<const bf16x8_t*>(smem + (ODD ? pb1 : pb0) + f * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 7; ++i) Af[i] = *reinterpret_cast<const bf16x8_t*>(smem + (ODD ? pa1 : pa0) + i * 2048);
    __builtin_amdgcn_sched_barrier(0);
    {
      const unsigned d = smem_base + f_off + (unsigned)(wave * 1024);
      if constexpr (!ODD) {
#pragma unroll
        for (int u = 0; u < 3; ++u) ntw_dma16(xr, d + (unsigned)(u * 8192), voffA[u], f_soff);
        if (grp == 0) ntw_dma16(xr, d + (unsigned)(3 * 8192), voffA[3], f_soff);
      } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) ntw_dma16(wr, d + (unsigned)(NTW_A_BYTES + u * 8192), voffB[u], f_soff);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!ODD) {
      if (tile_start) {   // the previous tile is complete (e_* still describe it); younger than its gelu_h loads: 3 + 4 + 4 pieces (3 + 3 + 3 in group 1)
        if (grp) ntw_wait_vm<9>(); else ntw_wait_vm<11>();
        epilogue();
        acc_zero();
      }
    } else {
      // this wave's pieces of the next step have landed: exactly this step's seven (4 X + 3 W; six in group 1) stay in flight
      if (grp) ntw_wait_vm<6>(); else ntw_wait_vm<7>();
      if constexpr (ACT == NTW_ACT_MULH) {
        if (c_ks == KS - 2) h_request();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (DBG == 5) tB = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 5) tC = __builtin_amdgcn_s_memtime();
    // ================= MFMA segment
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (i < 4 || nmw > i) {
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[f], Af[i], acc[i][f], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int m = i * 3 + f;
          if constexpr (ODD) {
            if (m == 0) {   // the compute stream moves on one step
              NTW_PIN_S(c_ks);
              ts_next = false;
              if (++c_ks == KS) {
                c_ks = 0;
                ++c_t;
                ts_next = true;
                if (c_t < my_tiles) {
                  decode(c_t, c_ms, c_hm, c_h0, c_col0);     // (e_* keep describing the finished tile until its epilogue has run)
                  nmw_nx = grp ? c_hm - c_h0 : c_h0;
                  a_cur = frag_lane + (unsigned)((grp ? c_h0 : 0) * 2048);
                }
              }
              NTW_PIN_S(c_ks);
            }
            if (m == 1) {
              NTW_PIN_S(r_off);
              r_off += (unsigned)NTW_SLOT;
              if (r_off == (unsigned)NTW_RING) r_off = 0;
              NTW_PIN_S(r_off);
            }
            if (m == 2) { pa0 = a_cur + r_off; NTW_PIN_V(pa0); }
            if (m == 3) { pa1 = pa0 ^ 64u; NTW_PIN_V(pa1); }
            if (m == 4) { pb0 = b_lane + r_off; NTW_PIN_V(pb0); }
            if (m == 5) { pb1 = pb0 ^ 64u; NTW_PIN_V(pb1); }
            if (m == 6) {
              NTW_PIN_S(f_off);
              f_off += (unsigned)NTW_SLOT;
              if (f_off == (unsigned)NTW_RING) f_off = 0;
              NTW_PIN_S(f_off);
            }
            if (m == 7) {   // the fill stream moves on one step
              NTW_PIN_S(f_ks);
              f_soff += 128u;
              if (++f_ks == KS) {
                f_ks = 0;
                f_soff = 0;
                ++f_t;
                int ms = 0, hm = 0, h0 = 0, col0 = 0;
                const bool live = f_t < my_tiles;
                if (live) decode(f_t, ms, hm, h0, col0);
                aim_fill(live, ms, hm, col0);
              }
              NTW_PIN_S(f_ks);
            }
          } else {
            if (m == 10) {   // the first phase of a tile: its epilogue addresses and bias (the previous tile's epilogue ran in this phase's issue segment)
              if (tile_start) e_load();
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (ODD) { tile_start = ts_next; nmw = nmw_nx; } else { tile_start = false; }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 5) tD = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG == 5) {
      const unsigned long long tE = __builtin_amdgcn_s_memtime();
      st_sum0 += tB - tA; st_sum1 += tC - tB; st_sum2 += tD - tC; st_sum3 += tE - tD; st_n += 1;
    }
  };
  const int total = my_tiles * KS;
  for (int s = 0; s < total; ++s) {
    phase(std::false_type{});
    phase(std::true_type{});
  }
#undef NTW_PIN_V
#undef NTW_PIN_S
  // the last tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  epilogue();
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SPG_DEV_KERNELS
  if constexpr (DBG == 5) {
    if (lane == 0 && blockIdx.x < 256) {
      unsigned long long* o = ntw_stamps + (blockIdx.x * 8 + wave) * 6;
      o[0] = st_sum0; o[1] = st_sum1; o[2] = st_sum2; o[3] = st_sum3; o[4] = st_n; o[5] = __builtin_amdgcn_s_memtime() - t_begin;
    }
  }
#endif
  (void)st_sum0; (void)st_sum1; (void)st_sum2; (void)st_sum3; (void)st_n; (void)t_begin;
}

// ---- host side ---------------------------------------------------------------------------------------------------
// Row blocks: heights of 8..14 fragments, the R whose (rounds x per-tile cost) is least; 0 = this kernel does not fit the problem
// (too few / too ragged tiles: the 128-row kernels of gemm.hip keep it).
static int ntw_pick_R(int M, int N, int K, int cus, float* util_out) {
  if (N % NTW_BN != 0 || K % 64 != 0 || K < 256) return 0;
  const int mf = (M + 15) / 16, T = N / NTW_BN;
  int best_R = 0;
  float best = 1e30f, best_util = 0.f;
  for (int R = (mf + NTW_MF - 1) / NTW_MF; R <= mf / 8; ++R) {
    const int hmax = (mf + R - 1) / R;
    if (hmax > NTW_MF || mf / R < 8) continue;
    const long tiles = (long)R * T;
    const long rounds = (tiles + cus - 1) / cus;
    const float cost = (float)rounds * ((float)(K / 64) * (float)(hmax + NTW_NF) + 0.5f * (float)hmax + 10.f);   // fill per step x steps + epilogue + fixed
    if (cost < best) { best = cost; best_R = R; best_util = (float)tiles / (float)(rounds * cus); }
  }
  if (util_out) *util_out = best_util;
  return best_R;
}

// 0 = launched, 1 = not applicable (the caller falls through), < 0 = error
int launch_nt_wide(const void* X, const void* W, void* C, const float* bias, const void* Hh, void* C2, int act, int M, int N, int K, int ldx,
                   int ldc, int cus, int dbg, hipStream_t s) {
  if (act != NTW_ACT_NONE && act != NTW_ACT_GELU && act != NTW_ACT_GELU_D && act != NTW_ACT_MULH) return 1;
  if (act == NTW_ACT_MULH && !Hh) return 1;
  if (act == NTW_ACT_GELU_D && !C2) return 1;
  if (act != NTW_ACT_MULH && Hh) return 1;
  if ((act == NTW_ACT_NONE || act == NTW_ACT_MULH) && C2) return 1;
  if (ldx % 8 != 0 || ldc % 8 != 0 || M < 1024) return 1;
  const long xb = (long)M * ldx * 2L, wb = (long)N * K * 2L, cb = ((long)(M - 1) * ldc + N) * 2L;
  if (xb >= 0x80000000L || wb >= 0x80000000L || cb >= 0x80000000L) return 1;
  float util = 0.f;
  const int R = ntw_pick_R(M, N, K, cus, &util);
  if (R == 0 || util < 0.8f) return 1;
  NtwArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.C2 = (bf16_t*)C2; a.Hh = (const bf16_t*)Hh; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldc = ldc;
  a.R = R; a.T = N / NTW_BN; a.mfrags = (M + 15) / 16; a.ntiles = R * a.T;
  a.xbytes = (unsigned)xb; a.wbytes = (unsigned)wb; a.cbytes = (unsigned)cb;
  const int grid = a.ntiles < cus ? a.ntiles : cus;
#define NTW_LAUNCH(A_, D_)                                                                                                     \
  do {                                                                                                                         \
    static bool attr_ = false;                                                                                                 \
    if (!attr_) {                                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_wide_kernel<A_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, NTW_RING); \
      attr_ = true;                                                                                                            \
    }                                                                                                                          \
    hipLaunchKernelGGL((gemm_nt_wide_kernel<A_, D_>), dim3(grid), dim3(512), NTW_RING, s, a);                                  \
  } while (0)
#ifdef SPG_DEV_KERNELS
  if (dbg == 5) {
    if (act == NTW_ACT_GELU_D) NTW_LAUNCH(NTW_ACT_GELU_D, 5);
    else if (act == NTW_ACT_MULH) NTW_LAUNCH(NTW_ACT_MULH, 5);
    else NTW_LAUNCH(NTW_ACT_NONE, 5);
    return check_launch("gemm_nt(wide, stamps)");
  }
#endif
  (void)dbg;
  if (act == NTW_ACT_GELU) NTW_LAUNCH(NTW_ACT_GELU, 0);
  else if (act == NTW_ACT_GELU_D) NTW_LAUNCH(NTW_ACT_GELU_D, 0);
  else if (act == NTW_ACT_MULH) NTW_LAUNCH(NTW_ACT_MULH, 0);
  else NTW_LAUNCH(NTW_ACT_NONE, 0);
#undef NTW_LAUNCH
  return check_launch("gemm_nt(wide)");
}

}  // namespace spg

extern "C" int spg_dev_ntw_stamps(unsigned long long* out) {   // 256 workgroups x 8 waves x 6 (SPG_NT_WIDE_DBG=5)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::ntw_stamps), sizeof(unsigned long long) * 256 * 8 * 6) == hipSuccess ? 0 : -1;
}
#endif  // SPG_DEV_KERNELS
