// Grouped weight-gradient GEMM for the wide trunk stages:  dW[N,K] += dY[M,N]^T X[M,K]  (+ dbias[N] += column sums of dY) for the
// four Linear layers of a Hiera block at once (reference: autograd of the nn.Linear layers of sam2's MultiScaleBlock, reached through
// models/feature_encoding.py:236), when every N and K is a multiple of 192 (stage 3: 576 / 1728 / 2304, stage 4: 1152 / 3456 / 4608).
//
// Why a second kernel beside gemm_tn_group4_kernel (gemm.hip): that one is bound by operand re-reads -- every 128 x 128 tile streams its
// own dY and X panels (573 MB through L2 per stage-3 block, 2.8x the algorithmic bytes, profiles/round2_bench_kernel_stats.md) -- and by
// its per-workgroup partial tiles.  Here, with the structure of the convolution weight gradient in conv_halo.hip:
//   * a workgroup owns ONE 256 x 192 block of some dW (the 192 side along the layer's 576-multiple dimension: 84 blocks for a stage-3
//     block against 243 tiles, 3.7 % padding) and a contiguous range of M; its 96 accumulator registers per lane live across the whole
//     range, one partial block per workgroup goes to a slab at the end and a small kernel adds the slabs in a fixed order
//     (deterministic).  Panel traffic per block drops to 0.58x.
//   * K (= M here) advances in 32-row slices: seven [32][64] bf16 sub-tiles (4 dY + 3 X or 3 + 4), 28 KiB, in a 5-slot ring (three slices
//     in flight behind the one being read); fragments by ds_read_b64_tr_b16 from 128-byte rows with the conflict-free column key of
//     conv_halo.hip; MFMA roles A = X (rows k), B = dY (cols n): D[k][n], a lane holds 4 consecutive k of one n.
//   * two wave groups one barrier apart: while one issues its 24-MFMA cluster the other issues the next slice's 20 transpose reads and
//     its share of the 28 LDS-DMA pieces; all per-slice address arithmetic sits in the gaps of the MFMA cluster.
// Two ways to fill the chip with such blocks:
//   * DIRECT (product, spg_gemm_tn_blocks): the wgrads of SEVERAL consecutive trunk blocks (stage 3: three of them = 252 blocks for 256 CUs)
//     go out as one launch in which every workgroup owns a whole block over ALL of M and adds its accumulators straight into dW / dbias:
//     no partial blocks, no slabs, no reduce launch, deterministic (one owner per gradient element).  The caller (models/engine.py) keeps
//     the dY / X operands of the deferred trunk blocks alive until the launch.
//   * split M (dev library only, SPG_TN_BLOCK=1: one trunk block per launch, S = 3 ranges of M per block, partial blocks through slabs
//     + a reduce launch).  Measured in round 3: main kernel 52.4 us against 65 for the tile kernel, but the 49 MB of partial blocks cost a
//     14.5 us reduce per trunk block: step unchanged (23.40 vs 23.39 ms), which is what the DIRECT form removes.
#include <algorithm>
#include <type_traits>
#include "common.h"

#ifndef TB_ABLATE
#define TB_ABLATE 0    // timing experiments with wrong results (tools/ builds only): 1 no MFMAs, 2 no fragment reads, 3 no LDS-DMA in the loop, 4 no dW read / store
#endif
#ifndef TB_MFMA32
#define TB_MFMA32 0    // 1 (A/B builds, -DTB_MFMA32=1): the DIRECT kernel on 32 x 32 x 16 MFMAs (tb_body32) -- measured no faster, see there
#endif
#ifndef TB_VARIANT
#define TB_VARIANT 0   // where the LDS-DMA pieces of slice sl + 3 are issued: 0 issue segment, 1 inside the MFMA cluster, 2 split, 3 = 1 without s_setprio
#endif

namespace spg {

typedef __attribute__((ext_vector_type(4))) unsigned tbrsrc_t;
typedef __attribute__((ext_vector_type(4))) short tbs16x4_t;
typedef __attribute__((ext_vector_type(8))) short tbs16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 tbbf16x2_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned tbbufvec_t;

__device__ __forceinline__ tbrsrc_t tb_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  tbrsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void tb_dma16(tbrsrc_t rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ int tb_key(int x) { return (x & 7) ^ (((x >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8_t tb_frag(const char* base, unsigned o0, unsigned o1) {
  const tbs16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tbs16x4_t*)(base + o0));
  const tbs16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tbs16x4_t*)(base + o1));
  const tbs16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

constexpr int TB_MAX_JOBS = 16;
constexpr int TB_SUB = 32 * 128;                 // one [32][64] bf16 sub-tile
constexpr int TB_SLOT = 7 * TB_SUB;              // one 32-row slice of a block's operands
constexpr int TB_NSLOT = 5;
constexpr int TB_LDS_BYTES = TB_NSLOT * TB_SLOT;
constexpr int TB_SLAB_FLOATS = 256 * 192;
constexpr unsigned TB_DEAD = 0x80000000u;

struct TbJob {
  const bf16_t* dY; const bf16_t* X; float* dW; float* dbias;
  int N, K, ldy, ldx, ldw;
  int wide_n;                // 1: block = 256 (n) x 192 (k); 0: 192 (n) x 256 (k)
  int tiles_b, tiles, tile0; // tiles along the 192 side, tiles of this job, first block id
};
struct TbGroup {
  TbJob job[TB_MAX_JOBS];
  int njobs, M, T, NB, S;    // T = 32-row slices of M, NB blocks in all, S splits of M per block
  // DIRECT form only.  overwrite: dW holds nothing worth keeping (the caller's gradients are zero at the start of a step and every block is
  // written exactly once): the block is stored, not added -- no cold read of the old values at the exit.  dbias is always added to (128 to
  // 256 floats per block; the qkv bias also receives the attention backward's padded-key gradient earlier in the step).  sq_part (or null): one
  // float per block = the sum of squares of everything the block's owner wrote (the FINAL gradient values, bias sums included), so that
  // the optimizer's global-norm clip need not read these gradients again (spg_sumsq_fold adds the partials in block order).
  int overwrite;
  float* sq_part;
};

// DIRECT: the workgroup owns the block over all of M (S == 1) and adds into dW / dbias itself; otherwise its partial block goes to `slab`
template <bool WIDE_N, bool DIRECT>
__device__ __forceinline__ void tb_body(const TbGroup& g, const TbJob& jb, int ta, int tb, int split, char* smem, float* __restrict__ slab,
                                        float* __restrict__ bslab, bool bias_blk, float* __restrict__ sq_out = nullptr) {
  constexpr int KA = WIDE_N ? 6 : 4;           // k blocks (16) per wave
  constexpr int NBk = WIDE_N ? 4 : 6;          // n blocks per wave
  constexpr int NYS = WIDE_N ? 4 : 3;          // dY sub-tiles of a slice (X: 7 - NYS)
  constexpr int NM = KA * NBk;                 // 24 MFMAs per slice
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wk = WIDE_N ? (wave & 1) : (wave & 3), wn = WIDE_N ? (wave >> 1) : (wave >> 2);
  const int q = lane >> 4, ra = (lane & 15) >> 2, rb = lane & 3;
  const int n0 = (WIDE_N ? ta * 256 : tb * 192), k0 = (WIDE_N ? tb * 192 : ta * 256);
  const int s0 = (int)((long)split * g.T / g.S), s1 = (int)((long)(split + 1) * g.T / g.S);
  const int nsl = s1 - s0;
  const tbrsrc_t yr = tb_rsrc(jb.dY, (unsigned)(((long)(g.M - 1) * jb.ldy + jb.N) * 2)), xr = tb_rsrc(jb.X, (unsigned)(((long)(g.M - 1) * jb.ldx + jb.K) * 2));
  const unsigned smem_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- fragment read offsets in slot 0: block-local column c lives in sub-tile c / 64 (dY sub-tiles first), 16-column group (c % 64) / 16
  unsigned xa[KA][2], ya[NBk][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 8 * q + ra + 4 * r;
#pragma unroll
    for (int kb = 0; kb < KA; ++kb) {
      const int col = wk * (KA * 16) + kb * 16;
      xa[kb][r] = (unsigned)((NYS + (col >> 6)) * TB_SUB + row * 128 + (((2 * ((col & 63) >> 4) + (rb >> 1)) ^ tb_key(row)) << 4) + (rb & 1) * 8);
    }
#pragma unroll
    for (int nb = 0; nb < NBk; ++nb) {
      const int col = wn * (NBk * 16) + nb * 16;
      ya[nb][r] = (unsigned)((col >> 6) * TB_SUB + row * 128 + (((2 * ((col & 63) >> 4) + (rb >> 1)) ^ tb_key(row)) << 4) + (rb & 1) * 8);
    }
  }
  // ---- LDS-DMA: the 28 pieces of a slice (7 sub-tiles x 4 pieces of 8 rows) are dealt pc = wave, wave + 8, ...: 4 pieces for waves 0-3, 3 for 4-7
  // piece pc: sub-tile pc >> 2, rows 8 (pc & 3) + lane / 8; a lane fetches logical chunk (lane & 7) ^ key(row)
  const int npc = wave < 4 ? 4 : 3;
  unsigned pbase[4];          // source offset of piece i for slice 0 of M (row part + column part), TB_DEAD when the columns are outside
  unsigned pstride[4];        // bytes per slice (32 rows)
  unsigned pdst[4];           // LDS offset inside a slot
  tbrsrc_t prs[4];            // ... and the descriptor it reads through (dY or X)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pc = wave + 8 * i;
    const int st = pc >> 2, row = 8 * (pc & 3) + (lane >> 3);
    const int chunk = (lane & 7) ^ tb_key(row);
    const bool isy = st < NYS;
    const int col = (isy ? n0 + 64 * st : k0 + 64 * (st - NYS)) + chunk * 8;
    const int ld = isy ? jb.ldy : jb.ldx;
    const bool inside = pc < 28 && col < (isy ? jb.N : jb.K);
    pbase[i] = inside ? (unsigned)((((long)s0 * 32 + row) * ld + col) * 2) : TB_DEAD;
    pstride[i] = inside ? (unsigned)(32 * ld * 2) : 0u;
    pdst[i] = (unsigned)(st * TB_SUB + (pc & 3) * 1024);
    prs[i] = isy ? yr : xr;
  }
  // (rows past M: beyond the descriptors' extent -> zero fill; slices past the range: never issued as live pieces)
  int is_slice = 0;            // slices issued so far
  unsigned is_slot = 0;        // ring slot (byte offset) the next issue fills
  auto issue_piece = [&](int i) __attribute__((always_inline)) {
    if (i < npc) {
      const unsigned off = is_slice < nsl ? pbase[i] : TB_DEAD;
      if (TB_ABLATE != 3 || is_slice < 3) tb_dma16(prs[i], smem_base + is_slot + pdst[i], off);
    }
  };
  auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(i);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    ++is_slice;
    is_slot = is_slot + TB_SLOT == (unsigned)TB_LDS_BYTES ? 0u : is_slot + TB_SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i) pbase[i] += pstride[i];
  };

  f32x4 acc[KA][NBk];
#pragma unroll
  for (int i = 0; i < KA; ++i)
#pragma unroll
    for (int j = 0; j < NBk; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[NBk];
#pragma unroll
  for (int j = 0; j < NBk; ++j) bsum[j] = 0.f;
  const tbbf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};
  const bool do_bias = (DIRECT ? bias_blk : bslab != nullptr) && wk == 0;

  // ---- prologue: three slices in flight, the first landed
  for (int i = 0; i < 3; ++i) { issue(); issue_advance(); }
  if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __syncthreads();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  bf16x8_t fx[KA], fy[NBk];
  unsigned rd = 0;             // ring slot being read (tracked inside xa / ya: they are advanced in place)
  for (int sl = 0; sl < nsl; ++sl) {
    // ================= issue segment: 20 transpose reads, this wave's pieces of slice sl + 3
    if (TB_ABLATE != 2 || sl == 0) {
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) fy[nb] = tb_frag(smem, ya[nb][0], ya[nb][1]);
#pragma unroll
      for (int kb = 0; kb < KA; ++kb) fx[kb] = tb_frag(smem, xa[kb][0], xa[kb][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#if TB_VARIANT == 0
    issue();
    __builtin_amdgcn_sched_barrier(0);
    // slice sl + 1 has landed (this wave's pieces: two younger slices may stay in flight); this wave's reads have returned
    if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#elif TB_VARIANT == 2
    issue_piece(0); issue_piece(1);
    __builtin_amdgcn_sched_barrier(0);
    // outstanding and allowed to stay: pieces 0, 1 of slice sl + 3 and all of slice sl + 2
    if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#else
    // the pieces of slice sl + 3 are issued inside the MFMA cluster below: here only slice sl + 2 may stay in flight
    if (wave < 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= MFMA segment; the ring / stream bookkeeping sits in its gaps
#if TB_VARIANT != 3
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int kb = m / NBk, nb = m % NBk;
      if (TB_ABLATE != 1 || sl == 0) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[kb], fy[nb], acc[kb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#if TB_VARIANT == 0
      constexpr int ADV = 0;
#else
      constexpr int ADV = 20;
      if (TB_VARIANT != 2 && m == 1) issue_piece(0);
      if (TB_VARIANT != 2 && m == 6) issue_piece(1);
      if (m == 11) issue_piece(2);
      if (m == 16) issue_piece(3);
#endif
      if (m == ADV) { asm volatile("" : "+s"(is_slice)); issue_advance(); asm volatile("" : "+s"(is_slice)); }
      if (m == ADV + 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pbase[i]));
      }
      if (m == 2) { rd = rd + TB_SLOT == (unsigned)TB_LDS_BYTES ? 0u : rd + TB_SLOT; asm volatile("" : "+s"(rd)); }
      if (m >= 3 && m < 3 + KA) {   // the fragment offsets follow the ring
        const int kb_ = m - 3;
        const unsigned d = rd == 0u ? (unsigned)-(TB_LDS_BYTES - TB_SLOT) : (unsigned)TB_SLOT;
        xa[kb_][0] += d; xa[kb_][1] += d;
        asm volatile("" : "+v"(xa[kb_][0])); asm volatile("" : "+v"(xa[kb_][1]));
      }
      if (m >= 3 + KA && m < 3 + KA + NBk) {
        const int nb_ = m - 3 - KA;
        const unsigned d = rd == 0u ? (unsigned)-(TB_LDS_BYTES - TB_SLOT) : (unsigned)TB_SLOT;
        ya[nb_][0] += d; ya[nb_][1] += d;
        asm volatile("" : "+v"(ya[nb_][0])); asm volatile("" : "+v"(ya[nb_][1]));
      }
      if (do_bias && m >= 13 && m < 13 + NBk) {   // column sums of dY from the fragments in registers
        const int nb_ = m - 13;
        const bf16x8_t v = fy[nb_];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const tbbf16x2_t pr = {v[2 * e], v[2 * e + 1]};
          bsum[nb_] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones2, bsum[nb_], false);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  constexpr int KE = WIDE_N ? 192 : 256;
  if constexpr (DIRECT) {
    // ---- the block is complete: dW[n0 + n][k0 + k] += acc.  A lane holds 4 consecutive k of one n, sixteen lanes sixteen different rows:
    // stored from there, every wave instruction touches 16 half lines (measured: 25 of 148 us per stage-3 launch, 74 of 202 at stage 4).
    // So the block goes through the (now idle) ring, half of its n rows at a time, and leaves as whole rows: consecutive lanes = consecutive
    // 16-byte pieces of a dW row (768 / 1024 contiguous bytes per row).
    constexpr int NE = WIDE_N ? 256 : 192;
    constexpr int RS = KE + 4;                     // floats per staged row (+4: the sixteen rows of a fragment store start 4 banks apart)
    constexpr int VR = KE / 4;                     // 16-byte pieces per row
    constexpr int NV = (NE / 2) * VR / 512;        // pieces per thread and half (12)
    static_assert((NE / 2) * RS * 4 <= TB_LDS_BYTES && (NE / 2) * VR % (512 * 6) == 0, "half a block fits the ring");
    float* stage = reinterpret_cast<float*>(smem);
    const bool keep_old = g.overwrite == 0;
    float ssq = 0.f;                               // sum of squares of the values this thread stores
    const int myhalf = WIDE_N ? (wn >> 1) : wn;
    const int nloc = (WIDE_N ? (wn & 1) * (NBk * 16) : 0) + (lane & 15);    // this lane's row inside its half, for nb = 0
    __syncthreads();                               // every wave's LDS-DMA has landed (vmcnt(0) above) and every fragment read is done
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (myhalf == h) {
#pragma unroll
        for (int kb = 0; kb < KA; ++kb)
#pragma unroll
          for (int nb = 0; nb < NBk; ++nb)
            *reinterpret_cast<f32x4*>(stage + (nloc + nb * 16) * RS + wk * (KA * 16) + kb * 16 + 4 * q) = acc[kb][nb];
      }
      __syncthreads();
      // six old values are requested before the first add (cold HBM lines: two latencies per half, not one per piece); only the loaded
      // values stay in registers, addresses are formed again for the stores (the other half's accumulators are still live)
      auto piece = [&](int it, int& row, int& c4) __attribute__((always_inline)) -> float* {
        const int v = it * 512 + tid;
        row = v / VR; c4 = v - row * VR;
        const int n = n0 + h * (NE / 2) + row, k = k0 + c4 * 4;
        return (n < jb.N && k < jb.K && TB_ABLATE != 4) ? jb.dW + (long)n * jb.ldw + k : nullptr;
      };
      constexpr int CH = 6;                        // pieces in flight per thread
#pragma unroll 1
      for (int c = 0; c < NV; c += CH) {
        f32x4 old[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          int row, c4;
          const float* d = piece(c + i, row, c4);
          old[i] = (d && keep_old) ? *reinterpret_cast<const f32x4*>(d) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          int row, c4;
          float* d = piece(c + i, row, c4);
          const f32x4 part = *reinterpret_cast<const f32x4*>(stage + row * RS + c4 * 4);
          if (d) {
            const f32x4 v = old[i] + part;
            *reinterpret_cast<f32x4*>(d) = v;
            ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
          }
        }
      }
      if (h == 0) __syncthreads();                 // the second half overwrites the staged rows
    }
    if (do_bias) {
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) {
        float b = bsum[nb];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        const int n = n0 + wn * (NBk * 16) + nb * 16 + (lane & 15);
        if (q == 0 && n < jb.N) {
          const float v = jb.dbias[n] + b;          // (always added: the qkv bias also receives the attention backward's padded-key gradient)
          jb.dbias[n] = v;
          ssq += v * v;
        }
      }
    }
    if (sq_out) {                                  // (wave-uniform) the block's sum of squares, summed in a fixed order
      __syncthreads();                             // every thread is done with the staged rows
      ssq = wave_sum(ssq);
      if (lane == 0) stage[wave] = ssq;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += stage[i];
        *sq_out = t;
      }
    }
  } else {
    // ---- the workgroup's partial block: slab[n local][k local] (rows of 192 or 256 floats: the reduce kernel knows the orientation)
#pragma unroll
    for (int kb = 0; kb < KA; ++kb)
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) {
        const int n = wn * (NBk * 16) + nb * 16 + (lane & 15), k = wk * (KA * 16) + kb * 16 + 4 * q;
        *reinterpret_cast<f32x4*>(slab + n * KE + k) = acc[kb][nb];
      }
    if (do_bias) {
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) {
        float b = bsum[nb];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) bslab[wn * (NBk * 16) + nb * 16 + (lane & 15)] = b;
      }
    }
  }
}

#if TB_MFMA32
// ---- the DIRECT body on 32 x 32 x 16 MFMAs: a measured experiment (A/B builds only) -------------------------------------------------
// RESULT: bit-checked by tests/test_kernels_gpu.py::test_gemm_tn_blocks, and no faster -- 142.5 vs 139.5 us per three stage-3 trunk
// blocks, 178.1 vs 177.3 at stage 4 (same box, tools/tn_blocks_bench.py): the partner's issue slots are not what bounds the interval.
// Same block, ring, LDS-DMA stream and two-group schedule as tb_body; what changes is the matrix instruction.  Why: in the ping-pong one
// group's issue segment (20 transpose reads, 3-4 LDS-DMA pieces, waits) runs beside the OTHER group's MFMA cluster on the same SIMD, and
// a 16 x 16 x 32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles, a 32 x 32 x 16 one for 8 of its 32 (MI355X_MICROARCH.md, cycle
// constants): the same FLOPs leave the partner 3/4 instead of 1/2 of the issue slots.  Ablations of the 16 x 16 body (tools builds,
// TB_ABLATE) had shown no single pole -- no MFMAs -39 us, no LDS-DMA -26, no reads -10 of 139 -- i.e. an issue-bound interval.
// Operands: A = X (32 k-columns x 16 rows of M), B = dY (16 rows of M x 32 n-columns); a lane holds column l % 32 and the 8 rows
// 8 (l / 32) .. + 7 of a 16-row k-step, so a 16-lane group q reads rows 16 s + 8 (q >> 1) + 4 r .. + 3 of column group (q & 1) with two
// ds_read_b64_tr_b16.  A 32-lane half of such a read touches 4 rows x 4 chunks; rows two apart share their banks, so the chunk key is
// ((row >> 1) & 1) << 2 (quad of chunks flipped every two rows): 64 distinct banks per half.
// D = 32 (k) x 32 (n): register i of a lane is k = 8 (i / 4) + 4 (l / 32) + i % 4 at n = l % 32.
typedef __attribute__((ext_vector_type(16))) float tbf32x16;
__device__ __forceinline__ int tb_key32(int row) { return ((row >> 1) & 1) << 2; }

template <bool WIDE_N>
__device__ __forceinline__ void tb_body32(const TbGroup& g, const TbJob& jb, int ta, int tb, char* smem, bool bias_blk) {
  constexpr int KA = WIDE_N ? 3 : 2;           // k blocks (32) per wave
  constexpr int NBk = WIDE_N ? 2 : 3;          // n blocks (32) per wave
  constexpr int NYS = WIDE_N ? 4 : 3;          // dY sub-tiles of a slice (X: 7 - NYS)
  constexpr int NM = 2 * KA * NBk;             // 12 MFMAs per slice (two 16-row k-steps)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wk = WIDE_N ? (wave & 1) : (wave & 3), wn = WIDE_N ? (wave >> 1) : (wave >> 2);
  const int q = lane >> 4, ra = (lane & 15) >> 2, rb = lane & 3;
  const int n0 = (WIDE_N ? ta * 256 : tb * 192), k0 = (WIDE_N ? tb * 192 : ta * 256);
  const int nsl = g.T;
  const tbrsrc_t yr = tb_rsrc(jb.dY, (unsigned)(((long)(g.M - 1) * jb.ldy + jb.N) * 2)), xr = tb_rsrc(jb.X, (unsigned)(((long)(g.M - 1) * jb.ldx + jb.K) * 2));
  const unsigned smem_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- fragment read offsets in slot 0, [k-step][block][r]
  unsigned xa[2][KA][2], ya[2][NBk][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = 16 * st + 8 * (q >> 1) + ra + 4 * r;
#pragma unroll
      for (int kb = 0; kb < KA; ++kb) {
        const int col = wk * (KA * 32) + kb * 32;
        xa[st][kb][r] = (unsigned)((NYS + (col >> 6)) * TB_SUB + row * 128 + (((2 * (((col & 63) >> 4) + (q & 1)) + (rb >> 1)) ^ tb_key32(row)) << 4) + (rb & 1) * 8);
      }
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) {
        const int col = wn * (NBk * 32) + nb * 32;
        ya[st][nb][r] = (unsigned)((col >> 6) * TB_SUB + row * 128 + (((2 * (((col & 63) >> 4) + (q & 1)) + (rb >> 1)) ^ tb_key32(row)) << 4) + (rb & 1) * 8);
      }
    }
  // ---- LDS-DMA: as in tb_body (28 pieces of 8 rows x 128 B per slice), with this body's chunk key
  const int npc = wave < 4 ? 4 : 3;
  unsigned pbase[4], pstride[4], pdst[4];
  tbrsrc_t prs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pc = wave + 8 * i;
    const int st = pc >> 2, row = 8 * (pc & 3) + (lane >> 3);
    const int chunk = (lane & 7) ^ tb_key32(row);
    const bool isy = st < NYS;
    const int col = (isy ? n0 + 64 * st : k0 + 64 * (st - NYS)) + chunk * 8;
    const int ld = isy ? jb.ldy : jb.ldx;
    const bool inside = pc < 28 && col < (isy ? jb.N : jb.K);
    pbase[i] = inside ? (unsigned)(((long)row * ld + col) * 2) : TB_DEAD;
    pstride[i] = inside ? (unsigned)(32 * ld * 2) : 0u;
    pdst[i] = (unsigned)(st * TB_SUB + (pc & 3) * 1024);
    prs[i] = isy ? yr : xr;
  }
  int is_slice = 0;
  unsigned is_slot = 0;
  auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < npc) {
        const unsigned off = is_slice < nsl ? pbase[i] : TB_DEAD;
        tb_dma16(prs[i], smem_base + is_slot + pdst[i], off);
      }
    }
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    ++is_slice;
    is_slot = is_slot + TB_SLOT == (unsigned)TB_LDS_BYTES ? 0u : is_slot + TB_SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i) pbase[i] += pstride[i];
  };

  tbf32x16 acc[KA][NBk];
#pragma unroll
  for (int i = 0; i < KA; ++i)
#pragma unroll
    for (int j = 0; j < NBk; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum[NBk];
#pragma unroll
  for (int j = 0; j < NBk; ++j) bsum[j] = 0.f;
  const tbbf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};
  const bool do_bias = bias_blk && wk == 0;

  // ---- prologue: three slices in flight, the first landed
  for (int i = 0; i < 3; ++i) { issue(); issue_advance(); }
  if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __syncthreads();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  bf16x8_t fx[2][KA], fy[2][NBk];
  unsigned rd = 0;
  for (int sl = 0; sl < nsl; ++sl) {
    // ================= issue segment: 20 transpose reads, this wave's pieces of slice sl + 3
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int nb = 0; nb < NBk; ++nb) fy[st][nb] = tb_frag(smem, ya[st][nb][0], ya[st][nb][1]);
#pragma unroll
      for (int kb = 0; kb < KA; ++kb) fx[st][kb] = tb_frag(smem, xa[st][kb][0], xa[st][kb][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue();
    __builtin_amdgcn_sched_barrier(0);
    // slice sl + 1 has landed (this wave's pieces: two younger slices may stay in flight); this wave's reads have returned
    if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= MFMA segment; the ring / stream bookkeeping sits in its gaps
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int st = m / (KA * NBk), kb = (m % (KA * NBk)) / NBk, nb = m % NBk;
      acc[kb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[st][kb], fy[st][nb], acc[kb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m == 0) { asm volatile("" : "+s"(is_slice)); issue_advance(); asm volatile("" : "+s"(is_slice)); }
      if (m == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pbase[i]));
        rd = rd + TB_SLOT == (unsigned)TB_LDS_BYTES ? 0u : rd + TB_SLOT; asm volatile("" : "+s"(rd));
      }
      if (m >= 2 && m < 2 + 2 * KA) {          // the fragment offsets follow the ring: one (k-step, block) pair of offsets per gap
        const int i_ = m - 2, st_ = i_ / KA, kb_ = i_ % KA;
        const unsigned d = rd == 0u ? (unsigned)-(TB_LDS_BYTES - TB_SLOT) : (unsigned)TB_SLOT;
        xa[st_][kb_][0] += d; xa[st_][kb_][1] += d;
        asm volatile("" : "+v"(xa[st_][kb_][0])); asm volatile("" : "+v"(xa[st_][kb_][1]));
      }
      if (m >= NM - 2 * NBk && m < NM) {       // (2 KA + 2 NBk = 10 pairs in gaps 2 .. 11)
        const int i_ = m - (NM - 2 * NBk), st_ = i_ / NBk, nb_ = i_ % NBk;
        const unsigned d = rd == 0u ? (unsigned)-(TB_LDS_BYTES - TB_SLOT) : (unsigned)TB_SLOT;
        ya[st_][nb_][0] += d; ya[st_][nb_][1] += d;
        asm volatile("" : "+v"(ya[st_][nb_][0])); asm volatile("" : "+v"(ya[st_][nb_][1]));
      }
      if (do_bias && m >= 2 && m < 2 + 2 * NBk) {   // column sums of dY from the fragments in registers
        const int i_ = m - 2, st_ = i_ / NBk, nb_ = i_ % NBk;
        const bf16x8_t v = fy[st_][nb_];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const tbbf16x2_t pr = {v[2 * e], v[2 * e + 1]};
          bsum[nb_] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones2, bsum[nb_], false);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- the block leaves through the idle ring as whole dW rows (see tb_body), half of its n rows at a time
  constexpr int KE = WIDE_N ? 192 : 256;
  constexpr int NE = WIDE_N ? 256 : 192;
  constexpr int RS = KE + 4;
  constexpr int VR = KE / 4;
  constexpr int NV = (NE / 2) * VR / 512;
  static_assert((NE / 2) * RS * 4 <= TB_LDS_BYTES && (NE / 2) * VR % (512 * 6) == 0, "half a block fits the ring");
  float* stage = reinterpret_cast<float*>(smem);
  const int myhalf = WIDE_N ? (wn >> 1) : wn;
  const int nloc = (WIDE_N ? (wn & 1) * (NBk * 32) : 0) + (lane & 31);    // this lane's row inside its half, for nb = 0
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (myhalf == h) {
#pragma unroll
      for (int kb = 0; kb < KA; ++kb)
#pragma unroll
        for (int nb = 0; nb < NBk; ++nb)
#pragma unroll
          for (int jq = 0; jq < 4; ++jq) {
            const f32x4 v = {acc[kb][nb][4 * jq], acc[kb][nb][4 * jq + 1], acc[kb][nb][4 * jq + 2], acc[kb][nb][4 * jq + 3]};
            *reinterpret_cast<f32x4*>(stage + (nloc + nb * 32) * RS + wk * (KA * 32) + kb * 32 + 8 * jq + 4 * (lane >> 5)) = v;
          }
    }
    __syncthreads();
    auto piece = [&](int it, int& row, int& c4) __attribute__((always_inline)) -> float* {
      const int v = it * 512 + tid;
      row = v / VR; c4 = v - row * VR;
      const int n = n0 + h * (NE / 2) + row, k = k0 + c4 * 4;
      return (n < jb.N && k < jb.K) ? jb.dW + (long)n * jb.ldw + k : nullptr;
    };
    constexpr int CH = 6;
#pragma unroll 1
    for (int c = 0; c < NV; c += CH) {
      f32x4 old[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        int row, c4;
        const float* d = piece(c + i, row, c4);
        old[i] = d ? *reinterpret_cast<const f32x4*>(d) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        int row, c4;
        float* d = piece(c + i, row, c4);
        const f32x4 part = *reinterpret_cast<const f32x4*>(stage + row * RS + c4 * 4);
        if (d) *reinterpret_cast<f32x4*>(d) = old[i] + part;
      }
    }
    if (h == 0) __syncthreads();
  }
  if (do_bias) {
#pragma unroll
    for (int nb = 0; nb < NBk; ++nb) {
      float b = bsum[nb];
      b += __shfl_xor(b, 32, 64);
      const int n = n0 + wn * (NBk * 32) + nb * 32 + (lane & 31);
      if (lane < 32 && n < jb.N) jb.dbias[n] += b;
    }
  }
}
#endif  // TB_MFMA32

__device__ __forceinline__ void tb_locate(const TbGroup& g, int blk, int& j, int& local) {
  j = 0;
#pragma unroll 1
  for (int i = 1; i < g.njobs; ++i) if (blk >= g.job[i].tile0) j = i;
  local = blk - g.job[j].tile0;
}

__device__ __forceinline__ int tb_xcd_order(int bid, int nwg) {
  const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + i;
}

// DIRECT launch: grid = min(blocks, CUs); a workgroup takes blocks wid, wid + grid, ... (one each when the set was sized to the chip)
__global__ __launch_bounds__(512) void tn_block_direct_kernel(TbGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // workgroups that share an XCD (blockIdx % 8: one L2) get a contiguous range of block ids: the ~32 workgroups of an XCD walk the SAME
  // rows of M through blocks of one or two layers, which share dY / X panels, so most of their fill is served by that L2
  const int wid = tb_xcd_order((int)blockIdx.x, (int)gridDim.x);
#pragma unroll 1
  for (int blk = wid; blk < g.NB; blk += (int)gridDim.x) {
    int j, local;
    tb_locate(g, blk, j, local);
    const TbJob& jb = g.job[j];
    const int ta = local / jb.tiles_b, tb = local - ta * jb.tiles_b;
    const bool bias_blk = jb.dbias != nullptr && (jb.wide_n ? tb == 0 : ta == 0);   // the blocks at the first k position of their n range
#if TB_MFMA32
    if (jb.wide_n) tb_body32<true>(g, jb, ta, tb, smem, bias_blk);
    else tb_body32<false>(g, jb, ta, tb, smem, bias_blk);
#else
    float* sq_out = g.sq_part ? g.sq_part + blk : nullptr;
    if (jb.wide_n) tb_body<true, true>(g, jb, ta, tb, 0, smem, nullptr, nullptr, bias_blk, sq_out);
    else tb_body<false, true>(g, jb, ta, tb, 0, smem, nullptr, nullptr, bias_blk, sq_out);
#endif
    __syncthreads();          // every wave has left the ring before the next block's prologue refills it
  }
}

#ifdef SPG_DEV_KERNELS
__global__ __launch_bounds__(512) void tn_block_kernel(TbGroup g, float* __restrict__ slabs, float* __restrict__ bslabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ids run block-fastest inside a split: the ~32 workgroups of an XCD walk the SAME rows of M through blocks that share dY / X panels
  // (a layer's 9 x 3 blocks read 12 panels, not 54)
  const int wid = tb_xcd_order((int)blockIdx.x, (int)gridDim.x);
  const int split = wid / g.NB, blk = wid - split * g.NB;
  const int wgid = blk * g.S + split;          // slab index: the reduce kernel walks a block's splits
  int j, local;
  tb_locate(g, blk, j, local);
  const TbJob& jb = g.job[j];
  const int ta = local / jb.tiles_b, tb = local - ta * jb.tiles_b;
  float* slab = slabs + (long)wgid * TB_SLAB_FLOATS;
  // the bias gradient: by the blocks at the first k position of their n range
  const bool bias_blk = jb.dbias != nullptr && (jb.wide_n ? tb == 0 : ta == 0);
  float* bslab = bias_blk ? bslabs + (long)wgid * 256 : nullptr;
  if (jb.wide_n) tb_body<true, false>(g, jb, ta, tb, split, smem, slab, bslab, bias_blk);
  else tb_body<false, false>(g, jb, ta, tb, split, smem, slab, bslab, bias_blk);
}

// dW[n0 + n][k0 + k] += sum over the S splits of block blk (split order); dbias likewise
__global__ __launch_bounds__(256) void tn_block_reduce_kernel(TbGroup g, const float* __restrict__ slabs, const float* __restrict__ bslabs) {
  const int blk = blockIdx.y;
  int j, local;
  tb_locate(g, blk, j, local);
  const TbJob& jb = g.job[j];
  const int ta = local / jb.tiles_b, tb = local - ta * jb.tiles_b;
  const int n0 = jb.wide_n ? ta * 256 : tb * 192, k0 = jb.wide_n ? tb * 192 : ta * 256;
  const int KE = jb.wide_n ? 192 : 256, NE = jb.wide_n ? 256 : 192;
  const int v = blockIdx.x * 256 + threadIdx.x;          // float4 index inside the block
  if (v < TB_SLAB_FLOATS / 4) {
    const int kq = KE / 4;
    const int n = v / kq, k = (v - n * kq) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = slabs + (long)blk * g.S * TB_SLAB_FLOATS + v * 4;
    for (int p = 0; p < g.S; ++p) s += *reinterpret_cast<const f32x4*>(base + (long)p * TB_SLAB_FLOATS);
    if (n0 + n < jb.N && k0 + k < jb.K && n < NE) {
      float* d = jb.dW + (long)(n0 + n) * jb.ldw + k0 + k;
      *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + s;
    }
  }
  if (jb.dbias && (jb.wide_n ? tb == 0 : ta == 0) && blockIdx.x == 0 && threadIdx.x < NE) {
    float b = 0.f;
    for (int p = 0; p < g.S; ++p) b += bslabs[((long)blk * g.S + p) * 256 + threadIdx.x];
    if (n0 + threadIdx.x < jb.N) jb.dbias[n0 + threadIdx.x] += b;
  }
}

// bytes of workspace the block kernel needs for `cus` workgroups (an upper bound for every problem set it accepts)
long tn_block_workspace_bytes(int cus) { return (long)cus * (TB_SLAB_FLOATS + 256) * 4L; }
#endif  // SPG_DEV_KERNELS

// fills g.job[] and returns the number of 256 x 192 blocks, or -1 when a problem is outside the kernel's domain
static long tb_plan(TbGroup& g, int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, int M, const int* N,
                    const int* K, const int* ldy, const int* ldx, const int* ldw) {
  if (njobs < 1 || njobs > TB_MAX_JOBS || M < 256) return -1;
  long nb = 0;
  for (int i = 0; i < njobs; ++i) {
    if (N[i] <= 0 || K[i] <= 0 || N[i] % 192 != 0 || K[i] % 192 != 0) return -1;
    if (ldy && (ldy[i] % 8 != 0 || ldx[i] % 8 != 0 || ldw[i] % 4 != 0 || ldy[i] < N[i] || ldx[i] < K[i] || ldw[i] < K[i])) return -1;
    if (ldy && ((long)M * ldy[i] * 2 >= 0x7FFFFFF0L || (long)M * ldx[i] * 2 >= 0x7FFFFFF0L)) return -1;
    TbJob& jb = g.job[i];
    jb.dY = dY ? (const bf16_t*)dY[i] : nullptr; jb.X = X ? (const bf16_t*)X[i] : nullptr; jb.dW = dW ? dW[i] : nullptr; jb.dbias = dbias ? dbias[i] : nullptr;
    jb.N = N[i]; jb.K = K[i]; jb.ldy = ldy ? ldy[i] : N[i]; jb.ldx = ldx ? ldx[i] : K[i]; jb.ldw = ldw ? ldw[i] : K[i];
    const long tn_ = (long)cdiv(N[i], 256) * (K[i] / 192), tk_ = (long)(N[i] / 192) * cdiv(K[i], 256);
    jb.wide_n = tn_ <= tk_ ? 1 : 0;            // the orientation with fewer (less padded) blocks
    jb.tiles_b = jb.wide_n ? K[i] / 192 : N[i] / 192;
    jb.tiles = (int)(jb.wide_n ? tn_ : tk_);
    jb.tile0 = (int)nb;
    nb += jb.tiles;
  }
  for (int i = njobs; i < TB_MAX_JOBS; ++i) g.job[i] = g.job[njobs - 1];
  return nb;
}

// number of 256 x 192 blocks the problems make (-1: outside the domain): the caller sizes its groups so that they just fill the CUs
long tn_blocks_count(int njobs, int M, const int* N, const int* K) {
  TbGroup g;
  return tb_plan(g, njobs, nullptr, nullptr, nullptr, nullptr, M, N, K, nullptr, nullptr, nullptr);
}

// DIRECT form; returns SPG_OK / an error, or 1 when the problem set is outside the domain.  More blocks than `cus`: several per workgroup,
// one after the other (the caller sizes its sets so that the rounds are full: models/engine.py)
int launch_tn_blocks_direct(int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, int M, const int* N,
                            const int* K, const int* ldy, const int* ldx, const int* ldw, int cus, hipStream_t s, int overwrite, float* sq_part) {
  TbGroup g;
  const long nb = tb_plan(g, njobs, dY, X, dW, dbias, M, N, K, ldy, ldx, ldw);
  if (nb < 1) return 1;
  g.njobs = njobs; g.M = M; g.T = cdiv(M, 32); g.NB = (int)nb; g.S = 1;
  g.overwrite = overwrite; g.sq_part = sq_part;
#if TB_MFMA32
  if (overwrite || sq_part) return 1;     // (the 32 x 32 x 16 A/B body keeps the plain exit)
#endif
  static bool attr_ = false;
  if (!attr_) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tn_block_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES);
    attr_ = true;
  }
  hipLaunchKernelGGL(tn_block_direct_kernel, dim3((unsigned)(nb < cus ? nb : cus)), dim3(512), TB_LDS_BYTES, s, g);
  return check_launch("tn_blocks(direct)");
}

#ifdef SPG_DEV_KERNELS
// split-M form: returns SPG_OK / an error, or 1 when the problem set is outside this kernel's domain (the caller uses gemm_tn_group4_kernel)
int launch_tn_block_group(int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, int M, const int* N,
                          const int* K, const int* ldy, const int* ldx, const int* ldw, void* workspace, long workspace_bytes, int cus,
                          hipStream_t s) {
  TbGroup g;
  const long nb = tb_plan(g, njobs, dY, X, dW, dbias, M, N, K, ldy, ldx, ldw);
  if (nb < 1 || nb > cus) return 1;             // (stage 4 at batch 8: 330 blocks of 18 slices pairs -- the tile kernel balances those better)
  const int T = cdiv(M, 32);
  int S = (int)(cus / nb);
  if (S > T / 16) S = T / 16;                   // at least 16 slices per workgroup
  if (S < 1) return 1;
  g.njobs = njobs; g.M = M; g.T = T; g.NB = (int)nb; g.S = S;
  g.overwrite = 0; g.sq_part = nullptr;
  const long need = nb * S * (long)(TB_SLAB_FLOATS + 256) * 4L;
  if (!workspace || workspace_bytes < need) return 1;
  float* slabs = (float*)workspace;
  float* bslabs = slabs + nb * S * (long)TB_SLAB_FLOATS;
  static bool attr_ = false;
  if (!attr_) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tn_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES);
    attr_ = true;
  }
  hipLaunchKernelGGL(tn_block_kernel, dim3((unsigned)(nb * S)), dim3(512), TB_LDS_BYTES, s, g, slabs, bslabs);
  int rc = check_launch("tn_block");
  if (rc) return rc;
  hipLaunchKernelGGL(tn_block_reduce_kernel, dim3(TB_SLAB_FLOATS / 4 / 256, (unsigned)nb), dim3(256), 0, s, g, (const float*)slabs, (const float*)bslabs);
  return check_launch("tn_block(reduce)");
}
#endif  // SPG_DEV_KERNELS

}  // namespace spg
