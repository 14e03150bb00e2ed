// 3x3 / pad 1 / stride 1 convolution over NHWC bf16 as an MFMA kernel with an LDS-RESIDENT INPUT TILE (forward and, with the flipped
// weight pack, dgrad).  Replaces nn.Conv2d(.., 3, padding=1) of the reference's EdgeDetectionModule / DecoderBlock
// (models/object_detection.py:115-123, 193-199, 230-236) on the bf16 path.
//
// Why not the implicit GEMM of gemm.hip: there every K step (one tap x 64 input channels) re-streams a 128-row X tile into LDS, so the
// LDS fill (64 FLOP per filled byte at a 128 x 128 tile) is what bounds it.  Here a workgroup owns an
// 8 x 32 pixel tile of one image and stages its (8+2) x (32+2) input halo ONCE per 64 input channels; the nine taps read their A
// fragments from that one image at shifted row addresses, and only the weights stream (BN x 128 B per step).  Fill per FLOP drops 3.5x
// (BN = 256) to 9x (BN = 64) against the 128 x 128 implicit GEMM.
//
// Geometry (cdna_hip_programming.md "The 256^2 8-phase template", re-derived for this data flow):
//   * workgroup = 8 waves = two GROUPS of four (waves 0-3 / 4-7: one wave of each group per SIMD).  The groups run the same program one
//     barrier apart: while one group's waves issue their 16-MFMA cluster, their SIMD partners issue the next phase's ds_reads and
//     LDS-DMA pieces -- the matrix pipe never waits for a wave's own loads.
//   * output tile 256 pixels x BN channels; BN = 256: waves 2(m) x 4(n), wave tile 128 x 64, 4 phases per step; BN = 128: 4 x 2,
//     64 x 64, 2 phases; BN = 64: 4 x 2, 64 x 32, 1 phase.  A phase = 4 m-blocks x 2 n-blocks x K 64 = 16 v_mfma_f32_16x16x32_bf16.
//   * LDS: two halo images of 44 KiB (340 rows of 128 B = one pixel x 64 channels, chunk c of halo column hx stored at c ^ (hx & 7);
//     +1 KiB dump piece), a 64 KiB weight ring (2 / 4 / 8 step slots), the bias vector.  155.7 KiB.
//   * weights: row r of a 32-row group holds W[n0 + 8 (r >> 2 & 3) + 4 (r >> 4) + (r & 3)], so the accumulators of n-blocks (2v, 2v+1)
//     hold 8 consecutive output channels of one pixel: 16-byte stores straight from the accumulators (the gemm_nt_v3 epilogue).
//   * MFMA roles: A operand = weights (n), B operand = pixels (m)  =>  D[n][m]: a lane holds 4 consecutive n of one pixel.
//
// Synchronisation (all waits are counted, nothing drains in the loop).  Phase g of group 0 spans barrier intervals 2g (reads + DMA
// issue) and 2g+1 (MFMA); group 1 runs one interval later.  A region of LDS that is read in phase g is re-filled by pieces issued in
// phase g+2 or later (every wave's reads have returned -- lgkmcnt(0) sits behind the barrier that follows the issue -- before the
// first piece can be issued), and is read again only in a phase that follows a phase in which EVERY wave has waited (vmcnt) for its
// own pieces of it (the barrier between the two phases publishes them).  vmcnt values are computed for the weight pieces alone;
// halo pieces and epilogue stores only add younger operations, which makes the waits stricter, never weaker.
#include <type_traits>
#include "common.h"

namespace spg {

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned hbufvec_t;
typedef __attribute__((ext_vector_type(4))) unsigned hrsrc_words_t;

__device__ __forceinline__ hrsrc_words_t halo_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  hrsrc_words_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
// LDS-DMA piece: 64 lanes x 16 B -> LDS[lds_addr .. +1024), lane l at +16 l; per-lane source offset, out-of-range offsets fill zeros.
// Inline asm: the compiler then tracks no LDS write in flight and guards no ds_read with vmcnt(0); ordering is this file's job.
__device__ __forceinline__ void halo_dma16(hrsrc_words_t rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
template <int N_> __device__ __forceinline__ void halo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int BN> struct HaloCfg;
template <> struct HaloCfg<256> { static constexpr int WM = 2, WN = 4, NW = 64, NPH = 4, NSLOT = 2, NG = 2, PPW = 2, LEAD = 2, VMW = 4; };
template <> struct HaloCfg<128> { static constexpr int WM = 4, WN = 2, NW = 64, NPH = 2, NSLOT = 4, NG = 2, PPW = 1, LEAD = 3, VMW = 5; };
template <> struct HaloCfg<64>  { static constexpr int WM = 4, WN = 2, NW = 32, NPH = 1, NSLOT = 8, NG = 1, PPW = 1, LEAD = 6, VMW = 5; };

constexpr int HALO_TH = 8, HALO_TW = 32;                    // output tile (rows x columns of pixels)
constexpr int HALO_PITCH = HALO_TW + 2;                     // halo columns per halo row
constexpr int HALO_ROWS = (HALO_TH + 2) * HALO_PITCH;       // 340 LDS rows of 128 B
constexpr int HALO_PIECES = (HALO_ROWS + 7) / 8;            // 43 pieces of 1 KiB
constexpr int HALO_BUF = (HALO_PIECES + 1) * 1024;          // + the dump piece
constexpr int HALO_WRING = 2 * HALO_BUF;                    // byte offset of the weight ring
constexpr int HALO_WRING_BYTES = 65536;
constexpr int HALO_BIAS = HALO_WRING + HALO_WRING_BYTES;    // f32 bias[Co] (Co <= 512)
constexpr int HALO_LDS_BYTES = HALO_BIAS + 2048;
constexpr unsigned HALO_DEAD = 0x80000000u;                 // source offset of a piece that must fill zeros (>= every buffer size)

#ifdef SPG_DEV_KERNELS
__device__ unsigned long long halo_stamps[256 * 8 * 6];   // DBG 5: per workgroup and wave: cycles in [issue | barrier 1 + lgkm wait | MFMA | barrier 2], phases, -
#endif

struct HaloArgs {
  const bf16_t* X; const bf16_t* Wp; bf16_t* C; const float* bias;
  int B, H, W, Ci, Co, ldc;
  int tiles_x, tiles_y, tiles_n, ntiles;
  unsigned xbytes, wbytes, cbytes;
};

// DBG (dev builds, wrong results by construction): 1 no LDS-DMA in the loop, 2 no fragment reads, 3 no MFMAs, 4 no epilogue stores
template <int BN, int DBG = 0>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(HaloArgs a) {
  using Cfg = HaloCfg<BN>;
  constexpr int WM = Cfg::WM, WN = Cfg::WN, NW = Cfg::NW, NPH = Cfg::NPH, NSLOT = Cfg::NSLOT, NG = Cfg::NG, PPW = Cfg::PPW, LEAD = Cfg::LEAD;
  constexpr int RPW = HALO_TH / WM;            // tile rows per wave
  constexpr int MB = RPW * 2;                  // 16-pixel m-blocks per wave
  constexpr int NBW = NW / 16;                 // n-blocks per wave
  constexpr int RG = WN * 32 * 128;            // bytes of one weight region (the rows every wave reads in one phase)
  constexpr int SLOT = NG * RG;                // bytes of one step's weight tile
  static_assert(SLOT * NSLOT == HALO_WRING_BYTES, "ring");
  static_assert(RG == 8 * PPW * 1024, "pieces per region");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = WM == 2 ? grp : (wave >> 1), wc = WM == 2 ? (wave & 3) : (wave & 1);
  const int r15 = lane & 15, q = lane >> 4;
  const int H = a.H, W = a.W, Ci = a.Ci;
  const int KC = Ci >> 6;
  const int G = (int)gridDim.x;
  // XCD-aware bijective remap: workgroups sharing blockIdx % 8 (one XCD, one L2) walk neighbouring tiles
  int first;
  {
    const int nwg = G, bid = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, i = bid >> 3;
    first = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + i;
  }
  if (first >= a.ntiles) return;
  const int my_tiles = (a.ntiles - first + G - 1) / G;
  const hrsrc_words_t xr = halo_rsrc_words(a.X, a.xbytes), wr = halo_rsrc_words(a.Wp, a.wbytes);
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, a.cbytes, 0x00020000);
  const unsigned smem_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem[0]

  // tile id -> (n tile, x tile, y tile, image)
  auto decode = [&](int t, int& n0, int& x0, int& y0, int& img) __attribute__((always_inline)) {
    const int nt = t % a.tiles_n, sp = t / a.tiles_n;
    const int xt = sp % a.tiles_x, r2 = sp / a.tiles_x;
    const int yt = r2 % a.tiles_y;
    img = r2 / a.tiles_y;
    n0 = nt * BN; x0 = xt * HALO_TW; y0 = yt * HALO_TH;
  };

  // ---- per-lane constants
  // A fragments: halo row p = (row + dyi) * 34 + 16 ch + dxi + r15 (row = tile row, dyi / dxi = tap + 1), chunk (4 s + q) ^ (hx & 7) with
  // hx = 16 ch + dxi + r15, i.e. (dxi + r15) & 7: the swizzle depends on the tap's dx only.
  unsigned a_lane[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) a_lane[d] = (unsigned)((d + r15) * 128 + ((q ^ ((d + r15) & 7)) << 4));
  // B fragments: row rho = wc * 32 + 16 b + r15 of a region, chunk (4 s + q) ^ (rho & 7)
  const unsigned b_lane = (unsigned)((wc * 32 + r15) * 128 + ((q ^ (r15 & 7)) << 4)) + (unsigned)HALO_WRING;
  // weight pieces: piece pi = wave * PPW + j of a region covers its rows 8 pi .. 8 pi + 7; lane -> row rho, chunk c
  const unsigned rowB = (unsigned)(9 * Ci * 2);
  unsigned w_rel[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int rho = 8 * (wave * PPW + j) + (lane >> 3);
    const int wc_ = rho >> 5, l = rho & 31, b = l >> 4, qq = (l >> 2) & 3, jj = l & 3;
    const int nrel = wc_ * NW + 8 * qq + 4 * b + jj;            // (+ 32 g for region g)
    w_rel[j] = (unsigned)nrel * rowB + (unsigned)((((lane & 7) ^ (rho & 7))) << 4);
  }

  // ---- stream states (wave-uniform)
  // compute stream
  int c_k = 0;                      // ordinal of the tile being multiplied
  int c_kc = 0, c_dyi = 0;          // 64-channel chunk, tap row (dy + 1)
  int c_slot = 0;                   // ring slot of the current step
  int c_hb = 0;                     // halo image of the current chunk
  // weight stream: LEAD steps ahead
  int w_k = 0, w_kc = 0, w_tap = 0, w_n0 = 0;
  bool w_live = true;
  // halo stream: one chunk ahead
  int h_img = 0, h_y0 = 0, h_x0 = 0, h_kc = 0;
  bool h_live = false;
  // epilogue parameters of the tile being multiplied
  unsigned e_base = 0;
  bool e_xok0 = false, e_xok1 = false;
  int e_y0 = 0, e_n0 = 0;

  auto w_soff = [&]() __attribute__((always_inline)) -> unsigned {
    return w_live ? (unsigned)(((w_n0 * 9 + w_tap) * Ci + w_kc * 64) * 2) : HALO_DEAD;
  };
  auto w_advance = [&]() __attribute__((always_inline)) {
    if (++w_tap == 9) {
      w_tap = 0;
      if (++w_kc == KC) {
        w_kc = 0;
        ++w_k;
        if (w_k < my_tiles) {
          if (a.tiles_n > 1) w_n0 = ((first + w_k * G) % a.tiles_n) * BN;
        } else {
          w_live = false;
        }
      }
    }
  };
  // the PPW pieces of region g of the weight stream's current step, into ring slot `slot`
  bool in_loop = false;
  auto w_issue = [&](int g, int slot) __attribute__((always_inline)) {
    if (DBG == 1 && in_loop) return;
    const unsigned so = w_soff() + (unsigned)(g * 32) * rowB;
    const unsigned dst = smem_base + (unsigned)(HALO_WRING + slot * SLOT + g * RG + wave * PPW * 1024);
#pragma unroll
    for (int j = 0; j < PPW; ++j) halo_dma16(wr, dst + j * 1024, w_live ? w_rel[j] + so : HALO_DEAD);
  };
  // piece k (0..5) of this wave of the halo stream's chunk, into halo image hb
  auto h_issue = [&](int k, int hb) __attribute__((always_inline)) {
    if (DBG == 1 && in_loop) return;
    const int pc = k * 8 + wave;
    const int p = pc * 8 + (lane >> 3);
    const int hy = (p * 241) >> 13, hx = p - hy * HALO_PITCH;
    const int y = h_y0 - 1 + hy, x = h_x0 - 1 + hx;
    const bool ok = h_live && p < HALO_ROWS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const unsigned off = (unsigned)((((h_img * H + y) * W + x) * Ci + h_kc * 64) * 2) + (unsigned)((((lane & 7) ^ (hx & 7))) << 4);
    const int pcd = pc < HALO_PIECES ? pc : HALO_PIECES;
    halo_dma16(xr, smem_base + (unsigned)(hb * HALO_BUF + pcd * 1024), ok ? off : HALO_DEAD);
  };
  auto e_load = [&](int t) __attribute__((always_inline)) {
    int n0, x0, y0, img;
    decode(t, n0, x0, y0, img);
    e_y0 = y0; e_n0 = n0;
    e_xok0 = x0 + r15 < W; e_xok1 = x0 + 16 + r15 < W;
    e_base = ((unsigned)((img * H + y0 + wm * RPW) * W + x0 + r15) * (unsigned)a.ldc + (unsigned)(n0 + wc * NW + 8 * q)) * 2u;
  };

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t Af[8], Bf[NG][4];

  // ---- epilogue of quadrant (m-half h, n-half g): + bias, bf16, 16-byte stores; the quadrant's accumulators restart at zero
  auto epilogue = [&](auto H_, auto G_) __attribute__((always_inline)) {
    constexpr int h = decltype(H_)::value, g = decltype(G_)::value;
    const float* bl = reinterpret_cast<const float*>(smem + HALO_BIAS) + e_n0 + wc * NW + g * 32 + 8 * q;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl), b1 = *reinterpret_cast<const f32x4*>(bl + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mi = 4 * h + i;
      const int row = 2 * h + (i >> 1);                                    // tile row inside the wave's rows
      const bool ok = ((i & 1) ? e_xok1 : e_xok0) && (e_y0 + wm * RPW + row < H);
      const unsigned off = e_base + (unsigned)(((row * W + 16 * (i & 1)) * a.ldc + g * 32) * 2);
      float ev[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ev[e] = acc[mi][2 * g][e] + b0[e]; ev[4 + e] = acc[mi][2 * g + 1][e] + b1[e]; }
      const u32x4 v = pack16<bf16_t>(ev);
      if constexpr (DBG != 4) __builtin_amdgcn_raw_buffer_store_b128(hbufvec_t{v.x, v.y, v.z, v.w}, cr, ok ? off : 0xFFFFFFF0u, 0, 0);
      else asm volatile("" ::"v"(v));
      acc[mi][2 * g] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[mi][2 * g + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // quadrant of phase p
  auto epilogue_of_phase = [&](auto P_) __attribute__((always_inline)) {
    constexpr int p = decltype(P_)::value;
    if constexpr (NPH == 4) {
      if constexpr (p == 0) epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if constexpr (p == 1) epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      if constexpr (p == 2) epilogue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      if constexpr (p == 3) epilogue(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    } else if constexpr (NPH == 2) {
      if constexpr (p == 0) epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if constexpr (p == 1) epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    } else {
      epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
  };

  // ---- prologue: bias -> LDS, halo image of chunk 0, weight steps 0 .. LEAD-1
  for (int i = tid; i < a.Co; i += 512) reinterpret_cast<float*>(smem + HALO_BIAS)[i] = a.bias ? a.bias[i] : 0.f;
  {
    int n0, x0, y0, img;
    decode(first, n0, x0, y0, img);
    w_n0 = n0;
    h_img = img; h_y0 = y0; h_x0 = x0; h_kc = 0; h_live = true;
    e_load(first);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) h_issue(k, 0);
#pragma unroll
  for (int s = 0; s < LEAD; ++s) {
#pragma unroll
    for (int g = 0; g < NG; ++g) w_issue(g, s);
    w_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  in_loop = true;
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  unsigned long long st_sum0 = 0, st_sum1 = 0, st_sum2 = 0, st_sum3 = 0, st_n = 0;
  const int total_it = my_tiles * KC * 3;              // iterations = (tile, chunk, tap row), three steps (dx) each
  bool prev_fin = false;                               // the previous step ended a tile: its last quadrant is still to be stored
  for (int it = 0; it < total_it; ++it) {
    const bool last_it = (c_kc == KC - 1) && (c_dyi == 2);
    if (c_dyi == 0) {   // a new chunk starts: aim the halo stream at the chunk after it
      if (c_kc + 1 < KC) {
        h_kc = c_kc + 1;          // same tile (h_img / h_y0 / h_x0 already describe it)
        h_live = true;
      } else if (c_k + 1 < my_tiles) {
        int n0;
        decode(first + (c_k + 1) * G, n0, h_x0, h_y0, h_img);
        h_kc = 0;
        h_live = true;
      } else {
        h_live = false;
      }
    }
    const unsigned sA = (unsigned)(c_hb * HALO_BUF + (wm * RPW + c_dyi) * (HALO_PITCH * 128));

    auto step = [&](auto DXI_) __attribute__((always_inline)) {
      constexpr int dxi = decltype(DXI_)::value;
      const unsigned a0 = a_lane[dxi] + sA, a1 = a0 ^ 64u;
      const unsigned b0 = b_lane + (unsigned)(c_slot * SLOT), b1 = b0 ^ 64u;
      const int sidx = c_dyi * 3 + dxi;                       // step of the chunk, 0..8
      const int w_slot = (c_slot + LEAD) & (NSLOT - 1);
      const bool fin = last_it && dxi == 2;

      auto read_A = [&](auto H_) __attribute__((always_inline)) {
        constexpr int h = decltype(H_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int imm = (2 * h + (i >> 1)) * (HALO_PITCH * 128) + (i & 1) * 2048;
          Af[2 * i] = *reinterpret_cast<const bf16x8_t*>(smem + a0 + imm);
          Af[2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(smem + a1 + imm);
        }
      };
      auto read_B = [&](auto G_) __attribute__((always_inline)) {
        constexpr int g = decltype(G_)::value;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          Bf[g][2 * b] = *reinterpret_cast<const bf16x8_t*>(smem + b0 + g * RG + b * 2048);
          Bf[g][2 * b + 1] = *reinterpret_cast<const bf16x8_t*>(smem + b1 + g * RG + b * 2048);
        }
      };
      auto mfmas = [&](auto H_, auto G_) __attribute__((always_inline)) {
        constexpr int h = decltype(H_)::value, g = decltype(G_)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[4 * h + i][2 * g + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[g][2 * b + s], Af[2 * i + s], acc[4 * h + i][2 * g + b], 0, 0, 0);
      };
      auto phase = [&](auto P_) __attribute__((always_inline)) {
        constexpr int p = decltype(P_)::value;
        constexpr int ph = NPH == 4 ? (p >= 2 ? 1 : 0) : 0;                     // m-half of this phase's quadrant
        constexpr int pg = NPH == 4 ? ((p == 1 || p == 2) ? 1 : 0) : (NPH == 2 ? p : 0);   // n-half
        unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
        if constexpr (DBG == 5) tA = __builtin_amdgcn_s_memtime();
        // ---- fragment reads
        if constexpr (DBG == 2) {
        } else if constexpr (NPH == 4) {
          if constexpr (p == 0) { read_B(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); read_A(std::integral_constant<int, 0>{}); }
          if constexpr (p == 1) read_B(std::integral_constant<int, 1>{});
          if constexpr (p == 2) read_A(std::integral_constant<int, 1>{});
        } else if constexpr (NPH == 2) {
          if constexpr (p == 0) { read_B(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); read_A(std::integral_constant<int, 0>{}); }
          if constexpr (p == 1) read_B(std::integral_constant<int, 1>{});
        } else {
          read_B(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); read_A(std::integral_constant<int, 0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LDS-DMA issue: the weight stream's region for this phase, one halo piece in steps 1..6 of a chunk
        if constexpr (NPH == 4) {
          if constexpr (p == 2) w_issue(0, w_slot);
          if constexpr (p == 3) { w_issue(1, w_slot); }
        } else if constexpr (NPH == 2) {
          w_issue(p, w_slot);
        } else {
          w_issue(0, w_slot);
        }
        if constexpr (p == 0) {
          if (sidx >= 1 && sidx <= 6) h_issue(sidx - 1, c_hb ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue pieces of a finished tile (in the other group's MFMA shadow)
        if constexpr (p == 0) {
          if constexpr (dxi == 0) {
            if (prev_fin) {
              epilogue_of_phase(std::integral_constant<int, NPH - 1>{});
              e_load(first + c_k * G);
            }
          }
        } else {
          if constexpr (dxi == 2) {
            if (fin) epilogue_of_phase(std::integral_constant<int, p - 1>{});
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the weight regions read in the next phase have landed (this wave's pieces; the barrier publishes everyone's)
        if constexpr (NPH == 4) {
          if constexpr (p == 3) halo_wait_vm<Cfg::VMW>();
        } else {
          halo_wait_vm<Cfg::VMW>();
        }
        if constexpr (DBG == 5) tB = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
#ifndef HALO_NO_LGKM0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG == 5) tC = __builtin_amdgcn_s_memtime();
#ifndef HALO_NO_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        if constexpr (DBG != 3) mfmas(std::integral_constant<int, ph>{}, std::integral_constant<int, pg>{});
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(Af[i]));
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(Bf[pg][i]));
        }
#ifndef HALO_NO_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG == 5) tD = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG == 5) {
          const unsigned long long tE = __builtin_amdgcn_s_memtime();
          st_sum0 += tB - tA; st_sum1 += tC - tB; st_sum2 += tD - tC; st_sum3 += tE - tD; st_n += 1;
        }
      };
      phase(std::integral_constant<int, 0>{});
      if constexpr (NPH >= 2) phase(std::integral_constant<int, 1>{});
      if constexpr (NPH == 4) { phase(std::integral_constant<int, 2>{}); phase(std::integral_constant<int, 3>{}); }
      // ---- advance one step
      w_advance();
      c_slot = (c_slot + 1) & (NSLOT - 1);
    };
    step(std::integral_constant<int, 0>{});
    prev_fin = false;
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    prev_fin = last_it;
    // ---- advance one iteration
    if (++c_dyi == 3) {
      c_dyi = 0;
      c_hb ^= 1;
      if (++c_kc == KC) { c_kc = 0; ++c_k; }
    }
  }
  // the last tile's last quadrant (c_k has moved past it: e_* still describe it)
  epilogue_of_phase(std::integral_constant<int, NPH - 1>{});
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SPG_DEV_KERNELS
  if constexpr (DBG == 5) {
    if (lane == 0 && blockIdx.x < 256) {
      unsigned long long* o = halo_stamps + (blockIdx.x * 8 + wave) * 6;
      o[0] = st_sum0; o[1] = st_sum1; o[2] = st_sum2; o[3] = st_sum3; o[4] = st_n; o[5] = 0;
    }
  }
#endif
  (void)st_sum0; (void)st_sum1; (void)st_sum2; (void)st_sum3; (void)st_n;
}

// returns SPG_OK, an error, or 1 when the problem is outside this kernel's domain (the caller falls back to the implicit GEMM)
int launch_conv3x3_halo(const void* X, const void* Wp, void* C, const float* bias, int B, int H, int W, int Ci, int Co, int ldc,
                        int cus, int force_bn, hipStream_t s) {
  const int dbg = force_bn / 1000;
  force_bn %= 1000;
  if (Ci % 64 != 0 || Co % 64 != 0 || Co > 512 || ldc % 8 != 0 || H < 1 || W < 1) return 1;
  const long M = (long)B * H * W;
  const long xb = M * Ci * 2L, wb = (long)Co * 9 * Ci * 2L, cb = ((M - 1) * ldc + Co) * 2L;
  if (xb >= 0x7FFFFFF0L || wb >= 0x7FFFFFF0L || cb >= 0xFFFFFFF0L) return 1;
  const int tx = cdiv(W, HALO_TW), ty = cdiv(H, HALO_TH);
  const long sp_tiles = (long)B * tx * ty;
  // tile width: the widest that divides Co, unless a narrower one fills the CUs' rounds markedly better
  int bn = Co % 256 == 0 ? 256 : (Co % 128 == 0 ? 128 : 64);
  auto util = [&](int b) {
    const long t = sp_tiles * (Co / b);
    return (double)t / (double)((t + cus - 1) / cus * cus);
  };
  if (bn == 256 && util(256) < 0.8 * util(128)) bn = 128;
  if (bn == 128 && util(128) < 0.8 * util(64)) bn = 64;
  if (force_bn == 256 || force_bn == 128 || force_bn == 64) {
    if (Co % force_bn != 0) return 1;
    bn = force_bn;
  }
  HaloArgs a;
  a.X = (const bf16_t*)X; a.Wp = (const bf16_t*)Wp; a.C = (bf16_t*)C; a.bias = bias;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.ldc = ldc;
  a.tiles_x = tx; a.tiles_y = ty; a.tiles_n = Co / bn;
  const long nt = sp_tiles * a.tiles_n;
  if (nt >= 0x7FFFFFFFL) return 1;
  a.ntiles = (int)nt;
  a.xbytes = (unsigned)xb; a.wbytes = (unsigned)wb; a.cbytes = (unsigned)cb;
  const int grid = a.ntiles < cus ? a.ntiles : cus;
#define SPG_HALO_LAUNCH(BN_, D_)                                                                                                    \
  do {                                                                                                                              \
    static bool attr_ = false;                                                                                                      \
    if (!attr_) {                                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                HALO_LDS_BYTES);                                                                                    \
      attr_ = true;                                                                                                                 \
    }                                                                                                                               \
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN_, D_>), dim3(grid), dim3(512), HALO_LDS_BYTES, s, a);                               \
  } while (0)
#define SPG_HALO_LAUNCH_BN(D_)                   \
  do {                                           \
    if (bn == 256) SPG_HALO_LAUNCH(256, D_);     \
    else if (bn == 128) SPG_HALO_LAUNCH(128, D_); \
    else SPG_HALO_LAUNCH(64, D_);                \
  } while (0)
#ifdef SPG_DEV_KERNELS
  if (dbg == 1) SPG_HALO_LAUNCH_BN(1);
  else if (dbg == 2) SPG_HALO_LAUNCH_BN(2);
  else if (dbg == 3) SPG_HALO_LAUNCH_BN(3);
  else if (dbg == 4) SPG_HALO_LAUNCH_BN(4);
  else if (dbg == 5) SPG_HALO_LAUNCH_BN(5);
  else
#endif
    SPG_HALO_LAUNCH_BN(0);
  (void)dbg;
#undef SPG_HALO_LAUNCH_BN
#undef SPG_HALO_LAUNCH
  return check_launch("conv3x3_halo");
}

}  // namespace spg

#ifdef SPG_DEV_KERNELS
extern "C" int spg_dev_halo_stamps(unsigned long long* out) {   // 256 workgroups x 8 waves x 6 (SPG_CONV_HALO_DBG=5)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::halo_stamps), sizeof(unsigned long long) * 256 * 8 * 6) == hipSuccess ? 0 : -1;
}
#endif
