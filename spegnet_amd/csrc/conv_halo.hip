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
#include <algorithm>
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
// WM x WN waves, NW output channels per wave, NPH phases per step (a phase = one m-half of the wave tile x all its n-blocks),
// NSLOT ring slots, LEAD = how many steps the weight stream runs ahead, VMW = vmcnt that retires the next step's weights
template <> struct HaloCfg<128> { static constexpr int WM = 4, WN = 2, NW = 64, NPH = 1, NSLOT = 4, LEAD = 3, VMW = 4; };
template <> struct HaloCfg<64>  { static constexpr int WM = 4, WN = 2, NW = 32, NPH = 1, NSLOT = 8, LEAD = 7, VMW = 6; };

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
__device__ unsigned long long halo_stamps[256 * 8 * 6];   // DBG 5: per workgroup and wave: cycles in [issue | barrier 1 | MFMA | barrier 2], phases, -
#endif

struct HaloArgs {
  const bf16_t* X; const bf16_t* Wp; bf16_t* C; const float* bias;
  float* stats;       // STATS instances: per (spatial tile, wave row) partial [2][Co] = sum and sum of squares of the bf16 outputs
  int B, H, W, Ci, Co, ldc;
  int tiles_x, tiles_y, tiles_n, ntiles;
  unsigned xbytes, wbytes, cbytes;
};

// DBG (dev builds, wrong results by construction): 1 no LDS-DMA in the loop, 2 no fragment reads, 3 no MFMAs, 4 no epilogue stores,
// 5 in-kernel stamps (tools/halo_stamps.py)
// STATS: the epilogue also forms the BatchNorm batch statistics of its tile (reference: nn.BatchNorm2d right behind every one of these
// convolutions, models/object_detection.py:119,194,198): per wave the column sums / sums of squares of the ROUNDED outputs over its 64
// pixels (in-lane over the four m-blocks, then a fixed 4-step DPP reduction over the 16 pixel lanes) go to row (spatial tile, wave row)
// of a partial matrix; spg_bn_stats_finalize_part sums the rows in a fixed order and finalises -- no second pass over the output.
__device__ __forceinline__ float halo_row_sum16(float v) {   // sum over the 16 lanes of a DPP row; the total lands in the row's lane 15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
  return v;
}

template <int BN, int DBG = 0, bool STATS = false>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(HaloArgs a) {
  using Cfg = HaloCfg<BN>;
  constexpr int WM = Cfg::WM, WN = Cfg::WN, NW = Cfg::NW, NSLOT = Cfg::NSLOT, LEAD = Cfg::LEAD;
  constexpr int RPW = HALO_TH / WM;            // tile rows per wave
  constexpr int MB = RPW * 2;                  // 16-pixel m-blocks per wave
  constexpr int NBW = NW / 16;                 // n-blocks per wave
  constexpr int SLOT = BN * 128;               // bytes of one step's weight tile
  constexpr int PPW = SLOT / 8192;             // weight pieces per wave and step
  constexpr int NM = MB * NBW * 2;             // MFMAs per step
  static_assert(SLOT * NSLOT == HALO_WRING_BYTES, "ring");
  static_assert(MB == 4 && NM >= 16, "the step hooks use gaps 0..15");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 1, wc = wave & 1;
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

  // ---- tile walk.  This workgroup multiplies tiles first, first + G, ...; a tile id is the mixed-radix number (image, y tile, x tile,
  // n tile), so "+ G" is an add with carries of G's digits: no division after this block.
  int gd_n, gd_x, gd_y, gd_i;          // digits of G
  int ed_n, ed_x, ed_y, ed_i;          // digits of the tile being multiplied (epilogue addresses, bias)
  int hd_n, hd_x, hd_y, hd_i;          // digits of the halo stream's tile (it runs one chunk ahead)
  int wd_n;                            // n digit of the weight stream's tile (it runs LEAD steps ahead; the lowest digit needs no carry in)
  {
    int r = G;
    gd_n = r % a.tiles_n; r /= a.tiles_n;
    gd_x = r % a.tiles_x; r /= a.tiles_x;
    gd_y = r % a.tiles_y; gd_i = r / a.tiles_y;
    r = first;
    ed_n = r % a.tiles_n; r /= a.tiles_n;
    ed_x = r % a.tiles_x; r /= a.tiles_x;
    ed_y = r % a.tiles_y; ed_i = r / a.tiles_y;
    hd_n = ed_n; hd_x = ed_x; hd_y = ed_y; hd_i = ed_i;
    wd_n = ed_n;
  }
  auto walk = [&](int& dn, int& dx, int& dy, int& di) __attribute__((always_inline)) {   // (dn, dx, dy, di) += G
    int v = dn + gd_n;
    int c = v >= a.tiles_n ? 1 : 0;
    dn = c ? v - a.tiles_n : v;
    v = dx + gd_x + c;
    c = v >= a.tiles_x ? 1 : 0;
    dx = c ? v - a.tiles_x : v;
    v = dy + gd_y + c;
    c = v >= a.tiles_y ? 1 : 0;
    dy = c ? v - a.tiles_y : v;
    di = di + gd_i + c;
  };

  // ---- per-lane constants
  // A fragments: halo row p = (row + dyi) * 34 + 16 ch + dxi + r15 (row = tile row, dyi / dxi = tap + 1), chunk (4 s + q) ^ (hx & 7) with
  // hx = 16 ch + dxi + r15, i.e. (dxi + r15) & 7: the swizzle depends on the tap's dx only.
  unsigned a_lane[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) a_lane[d] = (unsigned)((d + r15) * 128 + ((q ^ ((d + r15) & 7)) << 4)) + (unsigned)((wm * RPW) * (HALO_PITCH * 128));
  // B fragments: row rho = wc * 32 + 16 b + r15 of each 32-column half, chunk (4 s + q) ^ (rho & 7)
  // weight tile image: [n-half g][wave column wc][32 rows]: row = g * (WN * 32) + wc * 32 + 16 b + 4 qq + jj <-> n = wc NW + 32 g + 8 qq + 4 b + jj
  const unsigned b_lane = (unsigned)((wc * 32 + r15) * 128 + ((q ^ (r15 & 7)) << 4)) + (unsigned)HALO_WRING;
  constexpr int RG = WN * 32 * 128;            // bytes of one n-half of the weight tile
  const unsigned rowB = (unsigned)(9 * Ci * 2);
  const int ci2 = Ci * 2;
  // halo pieces: piece k (0..5) of this wave covers halo rows p = (8 k + wave) * 8 + lane / 8 = halo (row hy, column hx); per piece
  // the row, the column (out of range past the image's end) and the source offset relative to the halo's first pixel
  int hk_y[6], hk_x[6];
  unsigned hk_c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int p = (k * 8 + wave) * 8 + (lane >> 3);
    const int hy = p / HALO_PITCH, hx = p - hy * HALO_PITCH;
    hk_y[k] = hy;
    hk_x[k] = p < HALO_ROWS ? hx : 0x4000;
    hk_c[k] = (unsigned)((hy * W + hx) * ci2) + (unsigned)((((lane ^ hx) & 7)) << 4);
  }

  // ---- stream states (wave-uniform).  The nine taps of a chunk are unrolled, so a step's tap, the weight stream's tap and the halo
  // piece it issues are compile-time constants; what remains at run time is chunk-level state, advanced once per nine steps.
  const int nchunks = my_tiles * KC;
  const bool wg_stats = a.tiles_n == 1;   // STATS: one partial row per (workgroup, wave row) instead of per (tile, wave row)
  int c_k = 0, c_kc = 0;              // compute stream: tile ordinal and 64-channel chunk of the current chunk
  int c_hb = 0;                       // halo image the current chunk reads
  bool tile_start = false;            // the current chunk is the first of a tile other than the first (the previous tile's accumulators are pending)
  unsigned slot_off = 0;              // ring slot (byte offset) of the current step
  int w_k = 0, w_kc = 0, w_n0 = 0;    // weight stream (LEAD steps ahead): tile ordinal, chunk, first channel
  bool w_live = true;
  unsigned w_delta = 0;
  int h_y1 = 0, h_x1 = 0;             // halo stream (= the chunk after the compute stream's): its tile's origin - 1 (halo row / column 0)
  int h_sbase = 0;                    // byte offset of (image, h_y1, h_x1, channel 0); may be negative
  unsigned h_soff = HALO_DEAD;        // h_sbase + 128 * chunk, or DEAD when the stream has run out (every piece then fills zeros)
  unsigned e_base = 0;                // epilogue parameters of the tile being multiplied
  bool e_xok0 = false, e_xok1 = false;
  int e_y0 = 0;
  long e_prow = 0;                    // STATS: this wave's row of the partial-statistics matrix, as a float offset
  // values the issue segments of the current step use, prepared inside the previous step's MFMA cluster
  unsigned pa1 = 0, pb0 = 0, pb1 = 0, pw_off[PPW], ph_off = HALO_DEAD;
  unsigned pw_dst = 0, ph_dst = 0;
  bool in_loop = false;

  auto w_soff = [&](int tap) __attribute__((always_inline)) -> unsigned {
    return (unsigned)(((w_n0 * 9 + tap) * Ci + w_kc * 64) * 2);
  };
  auto e_load = [&]() __attribute__((always_inline)) {   // epilogue addresses of the tile being multiplied
    const int x0 = ed_x * HALO_TW, y0 = ed_y * HALO_TH, n0 = ed_n * BN;
    e_y0 = y0;
    e_xok0 = x0 + r15 < W; e_xok1 = x0 + 16 + r15 < W;
    e_base = ((unsigned)((ed_i * H + y0 + wm * RPW) * W + x0 + r15) * (unsigned)a.ldc + (unsigned)(n0 + wc * NW + 8 * q)) * 2u;
    if constexpr (STATS) {
      const long r_ = wg_stats ? (long)blockIdx.x : (long)((ed_i * a.tiles_y + ed_y) * a.tiles_x + ed_x);
      e_prow = (r_ * WM + wm) * (2L * a.Co) + n0 + wc * NW + 8 * q;
    }
  };
  auto h_aim = [&]() __attribute__((always_inline)) {    // the halo stream has entered tile hd
    h_y1 = hd_y * HALO_TH - 1; h_x1 = hd_x * HALO_TW - 1;
    h_sbase = ((hd_i * H + h_y1) * W + h_x1) * ci2;
  };
  auto h_off = [&](int k) __attribute__((always_inline)) -> unsigned {   // source offset of halo piece k (all at once: prologue only)
    const unsigned o = hk_c[k] + h_soff;
    return ((unsigned)(hk_y[k] + h_y1) < (unsigned)H && (unsigned)(hk_x[k] + h_x1) < (unsigned)W) ? o : HALO_DEAD;
  };

  f32x4 acc[MB][NBW];
  bf16x8_t Af[8], Bf[2 * NBW];
  // STATS: per-lane running sums / sums of squares of this lane's pixels for its 8 (NBW / 2) channels.  With one n-tile per pixel tile
  // (tiles_n == 1) they run over ALL tiles of the workgroup and are reduced and stored once, at the end: one partial row per (workgroup,
  // wave row); otherwise consecutive tiles cover different channels and every tile flushes its own row.
  float ssum[STATS ? NBW / 2 : 1][8], ssq[STATS ? NBW / 2 : 1][8];
  if constexpr (STATS) {
#pragma unroll
    for (int v = 0; v < NBW / 2; ++v)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ssum[v][e] = 0.f; ssq[v][e] = 0.f; }
  }
  auto stats_flush = [&](long prow) __attribute__((always_inline)) {
    if constexpr (STATS) {
#pragma unroll
      for (int v = 0; v < NBW / 2; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[v][e] = halo_row_sum16(ssum[v][e]); ssq[v][e] = halo_row_sum16(ssq[v][e]); }
      if (r15 == 15) {
#pragma unroll
        for (int v = 0; v < NBW / 2; ++v) {
          float* ps = a.stats + prow + v * 32;
          *reinterpret_cast<f32x4*>(ps) = f32x4{ssum[v][0], ssum[v][1], ssum[v][2], ssum[v][3]};
          *reinterpret_cast<f32x4*>(ps + 4) = f32x4{ssum[v][4], ssum[v][5], ssum[v][6], ssum[v][7]};
          *reinterpret_cast<f32x4*>(ps + a.Co) = f32x4{ssq[v][0], ssq[v][1], ssq[v][2], ssq[v][3]};
          *reinterpret_cast<f32x4*>(ps + a.Co + 4) = f32x4{ssq[v][4], ssq[v][5], ssq[v][6], ssq[v][7]};
        }
      }
#pragma unroll
      for (int v = 0; v < NBW / 2; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[v][e] = 0.f; ssq[v][e] = 0.f; }
    }
  };
  // the accumulators restart at the bias of the tile whose first channel is n0 (so the epilogue adds nothing)
  auto acc_init = [&](int n0) __attribute__((always_inline)) {
    const float* bl = reinterpret_cast<const float*>(smem + HALO_BIAS) + n0 + wc * NW + 8 * q;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      // n-block nb = 2 v + b holds channels 32 v + 8 q + 4 b + (0..3) of the wave's range in its four D rows 4 q + (0..3)
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 32 * (nb >> 1) + 4 * (nb & 1));
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i][nb] = bv;
    }
  };
  // ---- epilogue: bf16, 16-byte stores (n-blocks 2v, 2v+1 hold 8 consecutive channels of a pixel)
  auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const int row = mi >> 1;                                             // tile row inside the wave's rows
      const bool ok = ((mi & 1) ? e_xok1 : e_xok0) && (e_y0 + wm * RPW + row < H);
#pragma unroll
      for (int v = 0; v < NBW / 2; ++v) {
        const unsigned off = e_base + (unsigned)(((row * W + 16 * (mi & 1)) * a.ldc + v * 32) * 2);
        float ev[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ev[e] = acc[mi][2 * v][e]; ev[4 + e] = acc[mi][2 * v + 1][e]; }
        const u32x4 pk = pack16<bf16_t>(ev);
        if constexpr (DBG != 4) __builtin_amdgcn_raw_buffer_store_b128(hbufvec_t{pk.x, pk.y, pk.z, pk.w}, cr, ok ? off : 0xFFFFFFF0u, 0, 0);
        else asm volatile("" ::"v"(pk));
        if constexpr (STATS) {   // statistics of what was stored (the rounded values: what BatchNorm will normalise)
          float rv[8];
          unpack16<bf16_t>(pk, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x_ = ok ? rv[e] : 0.f;
            ssum[v][e] += x_;
            ssq[v][e] = __builtin_fmaf(x_, x_, ssq[v][e]);
          }
        }
      }
    }
    if constexpr (STATS) {
      if (!wg_stats) stats_flush(e_prow);
    }
  };

  // ---- prologue: bias -> LDS, halo image of chunk 0, weight steps 0 .. LEAD-1
  for (int i = tid; i < a.Co; i += 512) reinterpret_cast<float*>(smem + HALO_BIAS)[i] = a.bias ? a.bias[i] : 0.f;
  w_n0 = wd_n * BN;
  e_load();
  h_aim();
  h_soff = (unsigned)h_sbase;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int pc = k * 8 + wave;
    halo_dma16(xr, smem_base + (unsigned)((pc < HALO_PIECES ? pc : HALO_PIECES) * 1024), h_off(k));
  }
  static_assert(LEAD < 9, "the prologue's weight steps stay inside chunk 0");
  {
    // weight pieces: piece pi = wave * PPW + j covers image rows 8 pi .. 8 pi + 7; lane -> row rho, chunk c
    unsigned w_rel[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int rho = 8 * (wave * PPW + j) + (lane >> 3);
      const int g_ = rho / (WN * 32), r_ = rho % (WN * 32);
      const int wc_ = r_ >> 5, l = r_ & 31, b = l >> 4, qq = (l >> 2) & 3, jj = l & 3;
      const int nrel = wc_ * NW + 32 * g_ + 8 * qq + 4 * b + jj;
      w_rel[j] = (unsigned)nrel * rowB + (unsigned)((((lane & 7) ^ (rho & 7))) << 4);
    }
#pragma unroll
    for (int s = 0; s < LEAD; ++s) {
      const unsigned so = w_soff(s);
#pragma unroll
      for (int j = 0; j < PPW; ++j) halo_dma16(wr, smem_base + (unsigned)(HALO_WRING + s * SLOT + (wave * PPW + j) * 1024), w_rel[j] + so);
    }
    const unsigned so = w_soff(LEAD);
#pragma unroll
    for (int j = 0; j < PPW; ++j) pw_off[j] = w_rel[j] + so;     // (from here on the offsets move by scalar deltas)
  }
  // the halo stream moves on to chunk 1; step 0's prepared values
  if (KC > 1) {
    h_soff = (unsigned)(h_sbase + 128);
  } else if (my_tiles > 1) {
    walk(hd_n, hd_x, hd_y, hd_i);
    h_aim();
    h_soff = (unsigned)h_sbase;
  } else {
    h_soff = HALO_DEAD;
  }
  pw_dst = smem_base + (unsigned)(HALO_WRING + (LEAD & (NSLOT - 1)) * SLOT + wave * PPW * 1024);
  pa1 = a_lane[0] ^ 64u;
  pb0 = b_lane; pb1 = pb0 ^ 64u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc_init(ed_n * BN);
  in_loop = true;
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  unsigned long long st_sum0 = 0, st_sum1 = 0, st_sum2 = 0, st_sum3 = 0, st_n = 0;

#define HALO_PIN_V(x) asm volatile("" : "+v"(x))
#define HALO_PIN_S(x) asm volatile("" : "+s"(x))
  // ---- one step = tap T of the current chunk.  What the NEXT step's issue segment needs is formed in the gaps of this step's MFMA
  // cluster, one to three instructions per gap (VALU / SALU between a wave's own MFMAs is nearly free; in the issue segment it competes
  // with the partner wave's MFMAs and stretches the interval; a long block in ONE gap drains the matrix pipe).  Every piece is pinned to
  // its gap: its inputs and results pass through empty volatile asm statements, which the compiler keeps in order with the scheduling
  // fences (sched_barrier alone pins machine instructions, not where the arithmetic is materialised).
  auto step = [&](auto T_) __attribute__((always_inline)) {
    constexpr int t = decltype(T_)::value;
    constexpr int dyi = t / 3, dxi = t % 3;
    constexpr int nt = (t + 1) % 9, ndxi = nt % 3;          // the next step's tap
    constexpr int nwt = (nt + LEAD) % 9;                     // the weight stream's tap during the next step
    constexpr bool nh = nt >= 1 && nt <= 6;                  // the next step issues halo piece nt - 1
    constexpr int hk = nh ? nt - 1 : 0;
    int hv_y = 0, hv_x = 0;
    unsigned hv_o = 0;
    auto hook = [&](int m) __attribute__((always_inline)) {  // gap after MFMA m
      if (m == 0) {
        HALO_PIN_S(slot_off);
        slot_off = (slot_off + (unsigned)SLOT) & (unsigned)(HALO_WRING_BYTES - 1);
        HALO_PIN_S(slot_off);
      }
      if (m == 1) { pb0 = b_lane + slot_off; HALO_PIN_V(pb0); }
      if (m == 2) { pb1 = pb0 ^ 64u; HALO_PIN_V(pb1); }
      if (m == 3) {
        pw_dst = smem_base + (unsigned)HALO_WRING + ((slot_off + (unsigned)(LEAD * SLOT)) & (unsigned)(HALO_WRING_BYTES - 1)) + (unsigned)(wave * PPW * 1024);
        HALO_PIN_S(pw_dst);
      }
      if constexpr (nwt == 0) {   // the weight stream enters its next chunk
        if (m == 4) {   // scalar offset of (chunk, tap 0) minus that of (previous chunk, tap 8); a dead stream: everything out of range
          HALO_PIN_S(w_kc);
          const unsigned so_old = w_soff(8);
          if (++w_kc == KC) {
            w_kc = 0;
            ++w_k;
            if (w_k < my_tiles) {
              const int v = wd_n + gd_n;
              wd_n = v >= a.tiles_n ? v - a.tiles_n : v;
              w_n0 = wd_n * BN;
            } else {
              w_live = false;
            }
          }
          w_delta = w_soff(0) - so_old;
          HALO_PIN_S(w_delta);
        }
        if (m == 5) {
#pragma unroll
          for (int j = 0; j < PPW; ++j) { pw_off[j] = w_live ? pw_off[j] + w_delta : HALO_DEAD; HALO_PIN_V(pw_off[j]); }
        }
      } else {
        if (m == 4) {
#pragma unroll
          for (int j = 0; j < PPW; ++j) { HALO_PIN_V(pw_off[j]); pw_off[j] += (unsigned)ci2; HALO_PIN_V(pw_off[j]); }   // (a dead stream stays dead: DEAD + 8 tap strides does not wrap)
        }
      }
      if constexpr (nh) {   // the next step's halo piece
        if (m == 7) {
          const int pc = hk * 8 + wave;
          ph_dst = smem_base + (unsigned)((c_hb ^ 1) * HALO_BUF + (pc < HALO_PIECES ? pc : HALO_PIECES) * 1024);
          HALO_PIN_S(ph_dst);
        }
        if (m == 8) { hv_y = hk_y[hk] + h_y1; HALO_PIN_V(hv_y); }
        if (m == 9) { hv_x = hk_x[hk] + h_x1; HALO_PIN_V(hv_x); }
        if (m == 10) { hv_o = hk_c[hk] + h_soff; HALO_PIN_V(hv_o); }
        if (m == 11) { hv_o = (unsigned)hv_y < (unsigned)H ? hv_o : HALO_DEAD; HALO_PIN_V(hv_o); }
        if (m == 12) { ph_off = (unsigned)hv_x < (unsigned)W ? hv_o : HALO_DEAD; HALO_PIN_V(ph_off); }
      }
      if constexpr (t == 8) {   // the compute stream enters the next chunk; the halo stream the chunk after that
        if (m == 7) {
          HALO_PIN_S(c_kc);
          c_hb ^= 1;
          if (++c_kc == KC) { c_kc = 0; ++c_k; tile_start = true; walk(ed_n, ed_x, ed_y, ed_i); } else tile_start = false;
          HALO_PIN_S(c_kc);
        }
        if (m == 8) {
          const unsigned sa = c_hb ? (unsigned)HALO_BUF : (unsigned)-HALO_BUF;
#pragma unroll
          for (int d = 0; d < 3; ++d) { a_lane[d] += sa; HALO_PIN_V(a_lane[d]); }
        }
        if (m == 9) {
          if (c_kc + 1 < KC) {
            h_soff = c_k < my_tiles ? (unsigned)(h_sbase + (c_kc + 1) * 128) : HALO_DEAD;
          } else if (c_k + 1 < my_tiles) {
            walk(hd_n, hd_x, hd_y, hd_i);
            h_aim();
            h_soff = (unsigned)h_sbase;
          } else {
            h_soff = HALO_DEAD;
          }
          HALO_PIN_S(h_soff);
        }
      }
      if (m == 14) { pa1 = a_lane[ndxi] ^ 64u; HALO_PIN_V(pa1); }   // (after the halo image switch of t == 8)
      if constexpr (t == 0) {   // a new tile has begun (its predecessor's stores left in the issue segment): its epilogue addresses
        if (m == 13) { if (tile_start) e_load(); }
      }
    };
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
    if constexpr (DBG == 5) tA = __builtin_amdgcn_s_memtime();
    // ================= issue segment: fragment reads, LDS-DMA pieces, (at a tile's first step) the previous tile's stores
    if constexpr (DBG != 2) {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const int imm = (nb >> 1) * RG + (nb & 1) * 2048;
        Bf[2 * nb] = *reinterpret_cast<const bf16x8_t*>(smem + pb0 + imm);
        Bf[2 * nb + 1] = *reinterpret_cast<const bf16x8_t*>(smem + pb1 + imm);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int imm = ((i >> 1) + dyi) * (HALO_PITCH * 128) + (i & 1) * 2048;
        Af[2 * i] = *reinterpret_cast<const bf16x8_t*>(smem + a_lane[dxi] + imm);
        Af[2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(smem + pa1 + imm);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG == 1 && in_loop)) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) halo_dma16(wr, pw_dst + j * 1024, pw_off[j]);
      if constexpr (t >= 1 && t <= 6) halo_dma16(xr, ph_dst, ph_off);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t == 0) {
      if (tile_start) {   // the previous tile is complete; the accumulators restart at this tile's bias (ed already holds this tile's digits, the epilogue addresses still the previous tile's)
        epilogue();
        acc_init(ed_n * BN);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next step's weights have landed (this wave's pieces; the barrier publishes everyone's); this wave's reads have returned.
    // A chunk's last step also retires the next chunk's halo pieces: the youngest was issued in step 6, only the weight pieces of steps
    // 7 and 8 are younger.
    halo_wait_vm<(t == 8 && 2 * PPW < Cfg::VMW) ? 2 * PPW : Cfg::VMW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (DBG == 5) tB = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 5) tC = __builtin_amdgcn_s_memtime();
    // ================= MFMA segment
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int s = m / (MB * NBW), r = m % (MB * NBW), i = r / NBW, nb = r % NBW;
      if constexpr (DBG != 3) {
        acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[2 * nb + s], Af[2 * i + s], acc[i][nb], 0, 0, 0);
      } else {
        asm volatile("" ::"v"(Af[2 * i + s]), "v"(Bf[2 * nb + s]));
      }
      __builtin_amdgcn_sched_barrier(0);
      hook(m);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG == 5) tD = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG == 5) {
      const unsigned long long tE = __builtin_amdgcn_s_memtime();
      st_sum0 += tB - tA; st_sum1 += tC - tB; st_sum2 += tD - tC; st_sum3 += tE - tD; st_n += 1;
    }
  };
  for (int ch = 0; ch < nchunks; ++ch) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
  }
#undef HALO_PIN_V
#undef HALO_PIN_S
  // the last tile (the epilogue parameters still describe it)
  epilogue();
  if constexpr (STATS) {
    if (wg_stats) stats_flush(e_prow);
  }
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

// =====================================================================================================================
// Weight gradient of the 3x3 convolution on the same LDS-resident tiles:  dW[co][tap][ci] += sum_px dY[px][co] * X[px + tap][ci]
// (reference: autograd of nn.Conv2d, models/object_detection.py:115-123, 193-199).  A workgroup owns ONE 64 (co) x 9 x 64 (ci) block
// of dW and a contiguous range of 8 x 32 pixel tiles; its accumulators (72 VGPRs per lane) live across the whole range, so there is no
// per-tile epilogue at all: one partial block per workgroup goes to a slab at the end and a small kernel sums the slabs in a fixed
// order (deterministic: no float atomics, unlike the implicit-GEMM wgrad it replaces).  Per tile the workgroup stages dY [256 px][64 co]
// (32 KiB) and the X halo [340 px][64 ci] (43 KiB) ONCE for all nine taps -- the implicit GEMM re-gathers X per tap.
//   * contraction = pixels: a k32 step is one tile row of 32 pixels (k-step j <-> tile row j); tap (dy, dx) reads halo rows
//     (j + dy + 1) * 34 + dx + 1 + (0..31).  Both operands have k along LDS rows, so fragments come from ds_read_b64_tr_b16 (two per
//     fragment); the chunk swizzle key (x & 7) ^ (((x >> 3) & 1) << 2) of a pixel's column x makes those reads conflict-free.
//   * MFMA roles: A = X (rows ci), B = dY (cols co)  =>  D[ci][co]: a lane holds 4 consecutive ci of one co: 16-byte slab stores.
//   * waves: 2 (co halves of 32) x 4 (ci blocks of 16); a wave multiplies 2 co-blocks x 9 taps per k32 step: 11 fragments for 18 MFMAs.
//   * two wave groups one barrier apart, phases of two k-steps (36 MFMAs), as in the forward kernel; two tile buffers: the ten pieces per
//     wave of tile t + 1 are issued in phases 0 and 1 of tile t and retired (vmcnt(0)) in its phase 3, two phases after the last issue.
// =====================================================================================================================
typedef __attribute__((ext_vector_type(4))) short hs16x4_t;
typedef __attribute__((ext_vector_type(8))) short hs16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2_t;
constexpr int WG_DY_BYTES = 256 * 128;                       // dY tile image
constexpr int WG_BUF = WG_DY_BYTES + HALO_PIECES * 1024;     // + halo image: 76800 B
constexpr int WG_LDS_BYTES = 2 * WG_BUF;
constexpr int WG_SLAB_FLOATS = 64 * 9 * 64;

struct WgradArgs {
  const bf16_t* dY; const bf16_t* X; float* slabs; float* bslabs;
  int B, H, W, Ci, Co;
  int tiles_x, tiles_y, T, nci, P;      // spatial tiles in all, ci chunks, pixel-range splits per (co tile, ci chunk)
  unsigned ybytes, xbytes;
};

__device__ __forceinline__ int wg_key(int x) { return (x & 7) ^ (((x >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8_t wg_frag(const char* base, unsigned o0, unsigned o1) {
  const hs16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4_t*)(base + o0));
  const hs16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4_t*)(base + o1));
  const hs16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int DBG = 0>
__global__ __launch_bounds__(512) void conv3x3_wgrad_halo_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wco = wave & 1, wci = wave >> 1;           // this wave's co half (2 blocks of 16) and ci block
  const int q = lane >> 4, ra = (lane & 15) >> 2, rb = lane & 3;
  const int H = a.H, W = a.W, Ci = a.Ci, Co = a.Co;
  const int wg = blockIdx.x;
  const int combo = wg / a.P, split = wg - combo * a.P;
  const int co0 = (combo / a.nci) * 64, ci0 = (combo % a.nci) * 64;
  const int t0 = (int)((long)split * a.T / a.P), t1 = (int)((long)(split + 1) * a.T / a.P);
  const int ntiles = t1 - t0;
  const hrsrc_words_t yr = halo_rsrc_words(a.dY, a.ybytes), xr = halo_rsrc_words(a.X, a.xbytes);
  const unsigned smem_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- fragment read offsets (buffer 0; the buffer toggle is added per tile)
  unsigned xa[3][2], ya[2][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int hx = d + 8 * q + ra + 4 * r;
      xa[d][r] = (unsigned)(WG_DY_BYTES + hx * 128 + (((2 * wci + (rb >> 1)) ^ wg_key(hx)) << 4) + (rb & 1) * 8);
    }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int x = 8 * q + ra + 4 * r;
      ya[cb][r] = (unsigned)(x * 128 + (((2 * (2 * wco + cb) + (rb >> 1)) ^ wg_key(x)) << 4) + (rb & 1) * 8);
    }

  // ---- the prefetch stream's tile (one ahead of the multiplication) and the ten piece offsets prepared for it
  int pf_x, pf_y, pf_i, pf_left;        // tile digits of the tile the NEXT issue fills, tiles left to prefetch
  {
    int r = t0;
    pf_x = r % a.tiles_x; r /= a.tiles_x;
    pf_y = r % a.tiles_y; pf_i = r / a.tiles_y;
  }
  pf_left = ntiles;
  auto pf_advance = [&]() __attribute__((always_inline)) {
    --pf_left;
    if (++pf_x == a.tiles_x) { pf_x = 0; if (++pf_y == a.tiles_y) { pf_y = 0; ++pf_i; } }
  };
  unsigned po[10];
  // piece s (0..3: dY pieces 8 s + wave; 4..9: halo pieces 8 (s - 4) + wave) of the prefetch stream's tile
  auto piece_off = [&](int s) __attribute__((always_inline)) -> unsigned {
    const int l3 = lane >> 3;
    if (s < 4) {
      const int pc = s * 8 + wave;                       // px = 8 pc + l3: tile row pc >> 2, column (pc & 3) * 8 + l3
      const int cx = (pc & 3) * 8 + l3;
      const int y = pf_y * HALO_TH + (pc >> 2), x = pf_x * HALO_TW + cx;
      const bool ok = pf_left > 0 && y < H && x < W;
      const unsigned off = (unsigned)((((pf_i * H + y) * W + x) * Co + co0) * 2) + (unsigned)((((lane & 7) ^ wg_key(cx))) << 4);
      return ok ? off : HALO_DEAD;
    } else {
      const int pc = (s - 4) * 8 + wave;
      const int p = pc * 8 + l3;
      const int hy = (p * 241) >> 13, hx = p - hy * HALO_PITCH;
      const int y = pf_y * HALO_TH - 1 + hy, x = pf_x * HALO_TW - 1 + hx;
      const bool ok = pf_left > 0 && p < HALO_ROWS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      const unsigned off = (unsigned)((((pf_i * H + y) * W + x) * Ci + ci0) * 2) + (unsigned)((((lane & 7) ^ wg_key(hx))) << 4);
      return ok ? off : HALO_DEAD;
    }
  };
  auto piece_issue = [&](int s, int buf) __attribute__((always_inline)) {
    if (s < 4) {
      halo_dma16(yr, smem_base + (unsigned)(buf * WG_BUF + (s * 8 + wave) * 1024), po[s]);
    } else {
      const int pc = (s - 4) * 8 + wave;
      if (pc < HALO_PIECES) halo_dma16(xr, smem_base + (unsigned)(buf * WG_BUF + WG_DY_BYTES + pc * 1024), po[s]);
    }
  };

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float bsum[2] = {0.f, 0.f};
  const hbf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};
  const bool do_bias = a.bslabs != nullptr && ci0 == 0 && wci == 0;

  // ---- prologue: tile t0 into buffer 0; tile t0 + 1's offsets prepared
#pragma unroll
  for (int s = 0; s < 10; ++s) po[s] = piece_off(s);
#pragma unroll
  for (int s = 0; s < 10; ++s) piece_issue(s, 0);
  pf_advance();
#pragma unroll
  for (int s = 0; s < 10; ++s) po[s] = piece_off(s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier interval behind group 0

  bf16x8_t fx[2][9], fy[2][2];
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    auto phase = [&](auto PH_) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH_)::value;
      // ================= issue segment: 44 transpose reads (two k-steps), the next tile's pieces (phases 0 and 1)
      if constexpr (DBG != 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          constexpr int dummy = 0; (void)dummy;
          const int j = 2 * ph + ks;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) fy[ks][cb] = wg_frag(smem + j * 4096, ya[cb][0], ya[cb][1]);
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) fx[ks][tp] = wg_frag(smem + (j + tp / 3) * (HALO_PITCH * 128), xa[tp % 3][0], xa[tp % 3][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DBG != 1) {
        if constexpr (ph == 0) {
#pragma unroll
          for (int s = 0; s < 5; ++s) piece_issue(s, buf ^ 1);
        }
        if constexpr (ph == 1) {
#pragma unroll
          for (int s = 5; s < 10; ++s) piece_issue(s, buf ^ 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ph == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile has landed (this wave's pieces; the barrier publishes everyone's)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ================= MFMA segment; phases 2 and 3 prepare the piece offsets of the tile after next in its gaps
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < 36; ++m) {
        const int ks = m / 18, r = m % 18, tp = r >> 1, cb = r & 1;
        if constexpr (DBG != 3) acc[tp][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[ks][tp], fy[ks][cb], acc[tp][cb], 0, 0, 0);
        else asm volatile("" ::"v"(fx[ks][tp]), "v"(fy[ks][cb]));
        __builtin_amdgcn_sched_barrier(0);
        if (do_bias && m < 16) {   // column sums of dY (the bias gradient) from the fragments in registers: one element pair per gap
          const int ks_ = m >> 3, cb_ = (m >> 2) & 1, e = m & 3;
          const bf16x8_t v = fy[ks_][cb_];
          const hbf16x2_t pr = {v[2 * e], v[2 * e + 1]};
          bsum[cb_] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones2, bsum[cb_], false);
        }
        if constexpr (ph == 2) {
          if (m == 16) { asm volatile("" : "+s"(pf_left)); pf_advance(); asm volatile("" : "+s"(pf_left)); }
          if (m >= 18 && m < 28 && (m & 1) == 0) { const int s_ = (m - 18) >> 1; po[s_] = piece_off(s_); asm volatile("" : "+v"(po[s_])); }
        }
        if constexpr (ph == 3) {
          if (m >= 16 && m < 26 && (m & 1) == 0) { const int s_ = 5 + ((m - 16) >> 1); po[s_] = piece_off(s_); asm volatile("" : "+v"(po[s_])); }
          if (m == 28) {   // the next tile is multiplied from the other buffer
            const unsigned d = buf ? (unsigned)-WG_BUF : (unsigned)WG_BUF;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) { xa[dd][0] += d; xa[dd][1] += d; asm volatile("" : "+v"(xa[dd][0])); asm volatile("" : "+v"(xa[dd][1])); }
#pragma unroll
            for (int cb2 = 0; cb2 < 2; ++cb2) { ya[cb2][0] += d; ya[cb2][1] += d; asm volatile("" : "+v"(ya[cb2][0])); asm volatile("" : "+v"(ya[cb2][1])); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    buf ^= 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- the workgroup's partial block: slab[wg][co][tap][ci]
  float* sl = a.slabs + (long)wg * WG_SLAB_FLOATS;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int co = (2 * wco + cb) * 16 + (lane & 15), ci = wci * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(sl + ((co * 9 + tp) * 64 + ci)) = acc[tp][cb];
    }
  if (do_bias) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float b = bsum[cb];
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if (q == 0) a.bslabs[(long)wg * 64 + (2 * wco + cb) * 16 + (lane & 15)] = b;
    }
  }
}

// dW[co0 + co][tap][ci0 + ci] += sum over the P splits of a (co tile, ci chunk) in a FIXED order; dbias likewise from the ci chunk 0 slabs.
// A block owns 16 float4 of the 64 x 9 x 64 block; its 16 lanes-groups each sum every sixteenth split (up to four loads in flight per
// lane) and the sixteen partial sums are combined in a fixed tree: with P = 256 splits one thread per output walked 256 dependent-latency
// loads (55 us for 38 MB); 576 blocks x 256 threads stream them.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bslabs,
                                                                   float* __restrict__ dW, float* __restrict__ dbias, int Ci, int nci, int P,
                                                                   int torch_layout) {
  __shared__ f32x4 red[16][16];
  const int combo = blockIdx.y;
  const int co0 = (combo / nci) * 64, ci0 = (combo % nci) * 64;
  const int o = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int v = blockIdx.x * 16 + o;                     // float4 index inside the 64 x 9 x 64 block
  const float* base = slabs + (long)combo * P * WG_SLAB_FLOATS + v * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int p = pl;
  for (; p + 48 < P; p += 64) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(base + (long)p * WG_SLAB_FLOATS);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(base + (long)(p + 16) * WG_SLAB_FLOATS);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(base + (long)(p + 32) * WG_SLAB_FLOATS);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(base + (long)(p + 48) * WG_SLAB_FLOATS);
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; p < P; p += 16) s0 += *reinterpret_cast<const f32x4*>(base + (long)p * WG_SLAB_FLOATS);
  red[pl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0) {
    f32x4 t[4];
#pragma unroll
    for (int g_ = 0; g_ < 4; ++g_) t[g_] = (red[4 * g_][o] + red[4 * g_ + 1][o]) + (red[4 * g_ + 2][o] + red[4 * g_ + 3][o]);
    const f32x4 tt = (t[0] + t[1]) + (t[2] + t[3]);
    const int ci = (v & 15) * 4, tp = (v >> 4) % 9, co = (v >> 4) / 9;
    if (torch_layout) {   // dW is the parameter's own gradient [Co][Ci][3][3]: no packed scratch, no unpack launch
      float* d = dW + ((long)(co0 + co) * Ci + ci0 + ci) * 9 + tp;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[9 * e] += tt[e];
    } else {
      float* d = dW + ((long)(co0 + co) * 9 + tp) * Ci + ci0 + ci;
      *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + tt;
    }
  }
  if (dbias && bslabs && ci0 == 0 && blockIdx.x == 0) {   // (a serial walk over P = 256 splits took 65 us: four lanes per channel, four loads in flight)
    __shared__ float bred[4][64];
    const int c = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const float* bb = bslabs + (long)combo * P * 64 + c;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    int q_ = bl;
    for (; q_ + 12 < P; q_ += 16) {
      b0 += bb[(long)q_ * 64]; b1 += bb[(long)(q_ + 4) * 64]; b2 += bb[(long)(q_ + 8) * 64]; b3 += bb[(long)(q_ + 12) * 64];
    }
    for (; q_ < P; q_ += 4) b0 += bb[(long)q_ * 64];
    bred[bl][c] = (b0 + b1) + (b2 + b3);
    __syncthreads();
    if (bl == 0) dbias[co0 + c] += (bred[0][c] + bred[1][c]) + (bred[2][c] + bred[3][c]);
  }
}

// workspace floats of launch_conv3x3_wgrad_halo (0: the problem is outside its domain)
long conv3x3_wgrad_halo_ws_floats(int B, int H, int W, int Ci, int Co, int cus) {
  if (Ci % 64 != 0 || Co % 64 != 0 || H < 1 || W < 1) return 0;
  const long M = (long)B * H * W;
  if (M * Ci * 2L >= 0x7FFFFFF0L || M * Co * 2L >= 0x7FFFFFF0L) return 0;
  const int combos = (Co / 64) * (Ci / 64);
  if (combos > cus) return 0;
  const long T = (long)B * cdiv(W, HALO_TW) * cdiv(H, HALO_TH);
  long P = cus / combos;
  if (P > T / 4) P = T / 4;              // at least four tiles per workgroup (one or two are all prologue, slab write and reduce: EFE at 48 x 48 measured 54 vs 30 us)
  if (P < 1) return 0;
  if (T >= 64 && combos * P < cus / 2) return 0;   // fewer than half the CUs would work (EFE at 48 x 48: 35 vs 29 us for the implicit GEMM)
  return (long)combos * P * (WG_SLAB_FLOATS + 64);
}

// dW f32 [Co][9][Ci] += conv-wgrad(dY, X); dbias [Co] += column sums of dY (may be null).  Returns 1 when outside the kernel's domain.
int launch_conv3x3_wgrad_halo(const void* dY, const void* X, float* dW, float* dbias, float* ws, long ws_floats, int B, int H, int W, int Ci,
                              int Co, int cus, hipStream_t s, int torch_layout) {
  const long need = conv3x3_wgrad_halo_ws_floats(B, H, W, Ci, Co, cus);
  if (need <= 0 || !ws || ws_floats < need) return 1;
  WgradArgs a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.tiles_x = cdiv(W, HALO_TW); a.tiles_y = cdiv(H, HALO_TH);
  a.T = B * a.tiles_x * a.tiles_y;
  a.nci = Ci / 64;
  const int combos = (Co / 64) * a.nci;
  int P = cus / combos;
  if (P > a.T / 4) P = a.T / 4;
  a.P = P;
  a.slabs = ws;
  a.bslabs = dbias ? ws + (long)combos * P * WG_SLAB_FLOATS : nullptr;
  const long M = (long)B * H * W;
  a.ybytes = (unsigned)(M * Co * 2L); a.xbytes = (unsigned)(M * Ci * 2L);
  static bool attr_ = false;
  if (!attr_) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_halo_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES);
    attr_ = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<0>), dim3(combos * P), dim3(512), WG_LDS_BYTES, s, a);
  int rc = check_launch("conv3x3_wgrad_halo");
  if (rc) return rc;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(WG_SLAB_FLOATS / 4 / 16, combos), dim3(256), 0, s, (const float*)a.slabs,
                     (const float*)a.bslabs, dW, dbias, Ci, a.nci, P, torch_layout);
  return check_launch("conv3x3_wgrad_halo(reduce)");
}

// returns SPG_OK, an error, or 1 when the problem is outside this kernel's domain (the caller falls back to the implicit GEMM)
// stats != nullptr: also writes the BatchNorm partial statistics (conv3x3_halo_stats_rows() rows of 2 Co floats, every row written)
int launch_conv3x3_halo(const void* X, const void* Wp, void* C, const float* bias, int B, int H, int W, int Ci, int Co, int ldc,
                        int cus, int force_bn, hipStream_t s, float* stats) {
  const int dbg = force_bn / 1000;
  force_bn %= 1000;
  if (Ci % 64 != 0 || Co % 64 != 0 || Co > 512 || ldc % 8 != 0 || H < 1 || W < 1) return 1;
  const long M = (long)B * H * W;
  const long xb = M * Ci * 2L, wb = (long)Co * 9 * Ci * 2L, cb = ((M - 1) * ldc + Co) * 2L;
  if (wb >= 0x7FFFFFF0L) return 1;
  if (stats && (xb >= 0x7FFFFFF0L || cb >= 0xFFFFFFF0L)) return 1;
  if (xb >= 0x7FFFFFF0L || cb >= 0xFFFFFFF0L) {   // operands beyond one buffer descriptor's reach (batch 64 inference): images are independent, launch them in groups
    const long xi = (long)H * W * Ci * 2L, ci_ = (long)H * W * ldc * 2L;
    const long per = std::min(0x7FFFFFF0L / xi, 0xFFFFFFF0L / ci_);
    if (per < 1) return 1;
    for (long b0 = 0; b0 < B; b0 += per) {
      const int nb = (int)std::min(per, (long)B - b0);
      const int rc = launch_conv3x3_halo((const char*)X + b0 * xi, Wp, (char*)C + b0 * ci_, bias, nb, H, W, Ci, Co, ldc, cus, force_bn + 1000 * dbg, s, nullptr);
      if (rc != SPG_OK) return rc;     // (1 cannot happen here: the first group decides)
    }
    return SPG_OK;
  }
  const int tx = cdiv(W, HALO_TW), ty = cdiv(H, HALO_TH);
  const long sp_tiles = (long)B * tx * ty;
  // tile width: the widest that divides Co, unless a narrower one fills the CUs' rounds markedly better
  int bn = Co % 128 == 0 ? 128 : 64;
  auto util = [&](int b) {
    const long t = sp_tiles * (Co / b);
    return (double)t / (double)((t + cus - 1) / cus * cus);
  };
  if (bn == 128 && util(64) > 1.1 * util(128)) bn = 64;      // (measured: the 64-wide instance runs ~7 % behind at equal fill)
  if (stats && Co == 128) bn = 128;                            // one n-tile: the statistics stay in registers across the workgroup's tiles
  if (force_bn == 128 || force_bn == 64) {
    if (Co % force_bn != 0) return 1;
    bn = force_bn;
  }
  HaloArgs a;
  a.X = (const bf16_t*)X; a.Wp = (const bf16_t*)Wp; a.C = (bf16_t*)C; a.bias = bias; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.ldc = ldc;
  a.tiles_x = tx; a.tiles_y = ty; a.tiles_n = Co / bn;
  const long nt = sp_tiles * a.tiles_n;
  if (nt >= 0x7FFFFFFFL) return 1;
  a.ntiles = (int)nt;
  a.xbytes = (unsigned)xb; a.wbytes = (unsigned)wb; a.cbytes = (unsigned)cb;
  const int grid = a.ntiles < cus ? a.ntiles : cus;
#define SPG_HALO_LAUNCH(BN_, D_, S_)                                                                                                \
  do {                                                                                                                              \
    static bool attr_ = false;                                                                                                      \
    if (!attr_) {                                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN_, D_, S_>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS_BYTES);                                        \
      attr_ = true;                                                                                                                 \
    }                                                                                                                               \
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN_, D_, S_>), dim3(grid), dim3(512), HALO_LDS_BYTES, s, a);                           \
  } while (0)
#define SPG_HALO_LAUNCH_BN(D_)                                                              \
  do {                                                                                      \
    if (bn == 128) { if (stats) SPG_HALO_LAUNCH(128, D_, true); else SPG_HALO_LAUNCH(128, D_, false); } \
    else { if (stats) SPG_HALO_LAUNCH(64, D_, true); else SPG_HALO_LAUNCH(64, D_, false); }  \
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

// rows of the partial-statistics matrix a STATS launch writes (0: the problem is outside the kernel's domain)
long conv3x3_halo_stats_rows(int B, int H, int W, int Ci, int Co, int ldc, int cus) {
  if (Ci % 64 != 0 || Co % 64 != 0 || Co > 512 || ldc % 8 != 0 || H < 1 || W < 1) return 0;
  const long M = (long)B * H * W;
  if ((long)Co * 9 * Ci * 2L >= 0x7FFFFFF0L) return 0;
  if (M * Ci * 2L >= 0x7FFFFFF0L || ((M - 1) * ldc + Co) * 2L >= 0xFFFFFFF0L) return 0;   // (image groups: training batches stay below)
  const long sp = (long)B * cdiv(W, HALO_TW) * cdiv(H, HALO_TH);
  if (Co == 64 || Co == 128) return 4L * (sp < cus ? sp : cus);      // one n-tile: a row per (workgroup, wave row)
  return sp * 4L;                                                    // a row per (pixel tile, wave row)
}

}  // namespace spg

using namespace spg;
// ---- C ABI: bf16 3x3 convolution + BatchNorm partial statistics in one launch (include/spegnet_hip.h)
static int halo_cus(int cu_budget) {
  static int hw = 0;
  if (hw == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    hw = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return (cu_budget >= 8 && cu_budget < hw) ? cu_budget : hw;
}
extern "C" long spg_conv3x3_stats_rows(int dtype, int B, int H, int Wd, int Ci, int Co, int cu_budget) {
  return dtype == SPG_BF16 ? conv3x3_halo_stats_rows(B, H, Wd, Ci, Co, Co, halo_cus(cu_budget)) : 0;
}
extern "C" int spg_conv3x3_fwd_stats(int dtype, const void* X, const void* Wp, void* C, const float* bias, float* stats_part,
                                     long part_rows, int B, int H, int Wd, int Ci, int Co, int cu_budget, spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_BF16, "conv3x3_fwd_stats: bf16 only (dtype %d)", dtype);
  const int cus = halo_cus(cu_budget);
  const long rows = conv3x3_halo_stats_rows(B, H, Wd, Ci, Co, Co, cus);
  SPG_REQUIRE(rows > 0, "conv3x3_fwd_stats: no instance for B=%d H=%d W=%d Ci=%d Co=%d (ask spg_conv3x3_stats_rows first)", B, H, Wd, Ci, Co);
  SPG_REQUIRE(stats_part != nullptr && part_rows == rows, "conv3x3_fwd_stats: the partial matrix must have %ld rows of 2*Co floats, got %ld", rows, part_rows);
  const int rc = launch_conv3x3_halo(X, Wp, C, bias, B, H, Wd, Ci, Co, Co, cus, 0, (hipStream_t)stream, stats_part);
  if (rc == 1) { set_error("conv3x3_fwd_stats: problem outside the halo kernel's domain"); return SPG_ERR_UNSUPPORTED; }
  return rc;
}

// ---- C ABI: bf16 3x3 convolution weight gradient on LDS-resident tiles (include/spegnet_hip.h)
extern "C" long spg_conv3x3_wgrad_workspace_bytes(int dtype, int B, int H, int Wd, int Ci, int Co, int cu_budget) {
  return dtype == SPG_BF16 ? 4L * conv3x3_wgrad_halo_ws_floats(B, H, Wd, Ci, Co, halo_cus(cu_budget)) : 0;
}
extern "C" int spg_conv3x3_wgrad(int dtype, const void* dY, const void* X, float* dW, float* dbias, void* workspace, long workspace_bytes,
                                 int B, int H, int Wd, int Ci, int Co, int torch_layout, int cu_budget, spg_stream_t stream) {
  SPG_REQUIRE(dtype == SPG_BF16, "conv3x3_wgrad: bf16 only (dtype %d)", dtype);
  const int cus = halo_cus(cu_budget);
  const long need = 4L * conv3x3_wgrad_halo_ws_floats(B, H, Wd, Ci, Co, cus);
  SPG_REQUIRE(need > 0, "conv3x3_wgrad: no instance for B=%d H=%d W=%d Ci=%d Co=%d (ask spg_conv3x3_wgrad_workspace_bytes first)", B, H, Wd, Ci, Co);
  SPG_REQUIRE(workspace && workspace_bytes >= need, "conv3x3_wgrad: workspace of %ld bytes needed (got %ld)", need, workspace_bytes);
  const int rc = launch_conv3x3_wgrad_halo(dY, X, dW, dbias, (float*)workspace, workspace_bytes / 4, B, H, Wd, Ci, Co, cus, (hipStream_t)stream,
                                           torch_layout);
  if (rc == 1) { set_error("conv3x3_wgrad: problem outside the kernel's domain"); return SPG_ERR_UNSUPPORTED; }
  return rc;
}

#ifdef SPG_DEV_KERNELS
extern "C" int spg_dev_halo_stamps(unsigned long long* out) {   // 256 workgroups x 8 waves x 6 (SPG_CONV_HALO_DBG=5)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(spg::halo_stamps), sizeof(unsigned long long) * 256 * 8 * 6) == hipSuccess ? 0 : -1;
}
#endif
