/* libspegnet_hip.so -- C ABI of the MI355X-native SPEGNet hot path.
 *
 * The reference (Baber-Jan/SPEGNet) has no FFI / operator registry: its hot path is a chain of stock
 * torch.nn calls inside SPEGNet.forward (models/spegnet.py:137-206), the sam2 Hiera trunk
 * (models/feature_encoding.py:156-159,236), CODLoss (utils/loss_functions.py:242-295) and
 * Trainer._process_batch (engine/trainer.py:308-427).  Each entry point below replaces the group of
 * torch calls named in its comment.  See INTEGRATION.md for the ctypes binding.
 *
 * Conventions
 *   - dtype: SPG_F32 (parity path, exact-f32 MFMA) or SPG_BF16 (fast path, bf16 MFMA, f32 accumulate).
 *     "T" below means that storage type.  Parameters that stay fp32 in both modes are typed float*.
 *   - activations are NHWC (channels-last); encoder tokens are [B,H,W,C] == [M,C] row-major.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator); the library
 *     never allocates, frees or retains device memory, never synchronises, and launches only on
 *     `stream`, so every call is hipGraph-capturable and re-entrant (autograd thread safe).
 *   - return 0 on success, negative SPG_ERR_* otherwise; spg_last_error() gives the text.
 */
#ifndef SPEGNET_HIP_H
#define SPEGNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spg_stream_t; /* hipStream_t */

enum { SPG_F32 = 0, SPG_BF16 = 1 };
enum { SPG_OK = 0, SPG_ERR_BAD_ARG = -1, SPG_ERR_LAUNCH = -2, SPG_ERR_UNSUPPORTED = -3 };
enum { SPG_ACT_NONE = 0, SPG_ACT_GELU = 1, SPG_ACT_RELU = 2,
       /* gemm_nt, bf16: C = gelu(pre), C2 = gelu'(pre) -- the derivative is saved instead of the pre-activation (the MLP's backward
        * needs nothing else from it) -- and its partner for the backward GEMM: C = acc * gelu_h (+ residual), gelu_h = that derivative */
       SPG_ACT_GELU_SAVE_GRAD = 3, SPG_ACT_MUL_H = 4 };

/* ABI revision: bumped with every change of an exported signature.  Bindings must compare spg_version() with the SPG_ABI_VERSION they
 * were written against and refuse to run on a mismatch (spegnet_amd/_lib.py does).  300 = round 3. */
#define SPG_ABI_VERSION 309
int spg_version(void);
const char* spg_last_error(void);

/* ---- GEMM family (MFMA) ---------------------------------------------------------------------
 * spg_gemm_nt: C[M,N] = epi( X[M,K] . W[N,K]^T )      replaces F.linear / 1x1 conv / 3x3 conv fwd+dgrad
 *   epilogue order: +bias[n] (f32) -> store pre-activation to C2 (optional) -> act -> *gelu'(gelu_h[m,n])
 *   (optional) -> +residual[m,n] (optional) -> C.
 *   conv3x3 != 0: X is NHWC [B,H,W,Ci], M=B*H*W, K=9*Ci, k = tap*Ci+ci (pad 1, stride 1), W packed [N][9*Ci].
 * spg_gemm_tn: dW[N,K] += dY[M,N]^T . X[M,K]   (f32 atomic accumulation; same conv3x3 gather on X);
 *   dbias != NULL: also dbias[n] += sum_m dY[m][n] (the bias gradient, fused so dY is read once)
 *   replaces the weight-gradient half of linear/conv backward (engine/trainer.py:402 .backward()).
 * cu_budget (every GEMM entry point): number of CUs the persistent grid of THIS launch is sized for, 0 = all.  The kernels fill a
 *   CU completely, so a caller that overlaps them with another resident kernel (RCCL's all-reduce) leaves it some CUs instead of
 *   paying a second round.  A per-call argument: the library keeps no mutable state between calls.                               */
int spg_gemm_nt(int dtype, const void* X, const void* W, void* C, void* C2, const float* bias,
                const void* residual, const void* gelu_h, int M, int N, int K, int ldx, int ldc, int act,
                int conv3x3, int B, int H, int Wd, int Ci, int cu_budget, spg_stream_t stream);
int spg_gemm_tn(int dtype, const void* dY, const void* X, float* dW, float* dbias, void* workspace, long workspace_bytes,
                int M, int N, int K, int ldy, int ldx, int ldw, int conv3x3, int B, int H, int Wd, int Ci,
                int cu_budget, spg_stream_t stream);
/* bytes of caller-owned scratch that lets the M-splits of spg_gemm_tn write partial slabs with plain stores (+ one reduce
 * launch) instead of f32 atomics; 0 = not split.  workspace may be NULL (atomics are used).                             */
long spg_gemm_tn_workspace_bytes(int dtype, int M, int N, int K);
/* Up to 8 dense bf16 weight-gradient problems with the same row count M in one launch: dW[i][N,K] += dY[i][M,N]^T . X[i][M,K],
 * dbias[i][N] += colsum(dY[i]) (dbias, or single entries of it, may be NULL).  Every CU gets the same number of 64-row steps: whole
 * 128x128 tiles first (accumulated straight into dW), then an even share of the remaining tiles' steps, whose partial sums go
 * through the workspace (spg_gemm_tn_group_workspace_bytes(), caller-owned) and a small second kernel.  Pointer / int arrays are
 * HOST arrays of njobs entries.  Replaces the per-layer wgrad calls of a trunk block (reference: autograd of the four nn.Linear
 * of sam2's MultiScaleBlock).                                                                                                     */
int spg_gemm_tn_group(int dtype, int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias,
                      int M, const int* N, const int* K, const int* ldy, const int* ldx, const int* ldw, void* workspace,
                      long workspace_bytes, void* reduce_desc_out, int cu_budget, spg_stream_t stream);
/* reduce_desc_out != NULL (a HOST buffer of spg_gemm_tn_group_desc_bytes()): the second kernel is not launched; the caller keeps the
 * workspace and later folds up to 6 such launches at once with spg_gemm_tn_group_reduce_batch (descs / workspaces: HOST arrays of n
 * pointers: host descriptors, device workspaces).  The gradients are complete only after that call; a gradient buffer may appear
 * in only ONE pending launch (the batched fold adds into dW with plain read-modify-write).                                         */
long spg_gemm_tn_group_desc_bytes(void);
int spg_gemm_tn_group_reduce_batch(int n, const void* const* descs, const void* const* workspaces, spg_stream_t stream);
long spg_gemm_tn_group_workspace_bytes(void);
/* The weight gradients of SEVERAL trunk blocks in one launch (up to 16 dense bf16 problems sharing M, every N and K a multiple of 192):
 * each workgroup owns one whole 256 x 192 block of some dW over all of M and adds it straight into dW / dbias -- no partial sums, no
 * workspace, no second kernel, deterministic.  A set that makes more blocks than spg_num_cus(cu_budget) runs them in rounds (several
 * blocks per workgroup); spg_gemm_tn_blocks_count (host-only, -1 = outside the domain) tells the caller how many a set makes, so that
 * it can defer the wgrads of consecutive trunk blocks until their blocks fill whole rounds (stage 3 of Hiera-L: 84 blocks per trunk
 * block, three trunk blocks = 252 per launch; stage 4: 330 per trunk block, three = 990 = 3.87 rounds of 256).  Same reference op as spg_gemm_tn_group: autograd of the nn.Linear layers of sam2's MultiScaleBlock
 * (models/feature_encoding.py:236).  The caller keeps every dY / X alive until the launch.                                           */
/* overwrite != 0: dW is STORED, not added to (valid when it holds zeros -- the start of a step -- and no other launch of the step writes
 * it): the exit skips the cold read of the old values; dbias is always added to.  sq_part != NULL: f32 [spg_gemm_tn_blocks_count()] -- element i receives
 * the sum of squares of everything block i's owner wrote (the final gradient values of that block, bias sums included), so the optimizer's
 * global-norm clip (engine/trainer.py:404-406, clip_grad_norm_) need not read these gradients again: spg_sumsq_fold.                    */
int spg_gemm_tn_blocks(int dtype, int njobs, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias,
                       int M, const int* N, const int* K, const int* ldy, const int* ldx, const int* ldw, int overwrite, float* sq_part,
                       int cu_budget, spg_stream_t stream);
long spg_gemm_tn_blocks_count(int njobs, int M, const int* N, const int* K);
int spg_num_cus(int cu_budget);   /* CUs the persistent grids are sized for under this budget (0 = all) */
/* ---- weight packing (per optimizer step): f32 master -> T copies -----------------------------------
 * spg_pack_matrix: dst[r][c] = src[r][c] (transpose=0) or dst[c][r] = src[r][c] (transpose=1), src f32 [R,C].
 * spg_pack_conv3x3: torch [Co,Ci,3,3] f32 -> fwd pack [Co][tap][Ci] and dgrad pack [Ci][tap'][Co] (tap' flipped).
 * spg_unpack_conv3x3_grad: packed f32 grad [Co][tap][Ci] -> torch layout [Co,Ci,3,3] (accumulates: dst += packed). */
int spg_pack_matrix(int dtype, const float* src, void* dst, int R, int C, int transpose, spg_stream_t stream);
/* spg_pack_batch: the same for a whole table of matrices in ONE launch.  jobs = device array of
 * struct { const float* src; void* dst; void* dst_t; int R, C, tile0, pad; }  -- dst = [R][C] copy or NULL, dst_t = [C][R] or NULL
 * (tile0 = prefix sum of ceil(R/32)*ceil(C/32)); one read of the fp32 master produces both layouts.              */
int spg_pack_batch(int dtype, const void* jobs, int njobs, int total_tiles, spg_stream_t stream);
int spg_pack_conv3x3(int dtype, const float* src, void* dst_fwd, void* dst_dgrad, int Co, int Ci, spg_stream_t stream);
int spg_unpack_conv3x3_grad(const float* packed, float* dst, int Co, int Ci, spg_stream_t stream);

/* ---- Hiera trunk pieces (sam2 Hiera MultiScaleBlock) --------------------------------
 * layernorm: y = LN(x)*g+b over last dim C (eps), stats saved (mean,rstd f32 [M]) for backward.
 * layernorm_bwd: dx = LN'(dy) (+ dres if non-null); dgamma/dbeta (f32) += the column sums, by a deterministic reduction
 *   (see "Deterministic reductions" below for red_ws / red_counters; both may be NULL when dgamma == dbeta == NULL).   */
int spg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int M, int C, float eps, spg_stream_t stream);
int spg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int M, int C,
                      float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
/* the dgamma / dbeta part of layernorm_bwd for up to 48 jobs in one launch (they only feed the optimizer, so the trunk backward
 * defers them: 2 launches per step instead of 96).  A job is C <= 256 16-byte chunks of columns of a [M, ld] matrix (wider rows are
 * split by the caller).  Arrays are HOST arrays of njobs entries; dgamma / dbeta accumulate (+=).                                 */
int spg_layernorm_param_grads_batch(int dtype, int njobs, const void* const* dy, const void* const* x, const float* const* mean,
                                    const float* const* rstd, float* const* dgamma, float* const* dbeta, const int* M, const int* C,
                                    const int* ld, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
/* floats of red_ws the batched call needs for these column counts (HOST array); it also needs njobs zeroed counters */
long spg_layernorm_param_grads_batch_workspace_floats(int njobs, const int* C);
/* windowed multi-head attention straight from the fused qkv projection output [B,H,W,3,heads,hd].
 * Windows of ws x ws over the HxW map (ws<=0: one global window); padded window slots act as keys equal to
 * the qkv bias (what zero-padding after LN produces in the reference) with multiplicity folded in.
 * q_pooled != NULL: queries come from a 2x2-max-pooled map [B,H/2,W/2,heads*hd] (stage-transition blocks),
 * windows on the query side are ws/2.  lse f32 [B,Hq,Wq,heads] saved for backward.                        */
int spg_attn_fwd(int dtype, const void* qkv, const void* q_pooled, const void* qkv_bias_t, void* out, float* lse,
                 int B, int H, int W, int heads, int hd, int ws, spg_stream_t stream);
/* dqkv [B,H,W,3,heads,hd] (q part written only when q_pooled==NULL, else dq_pooled [B,H/2,W/2,heads*hd]);
 * dbias_pad f32 [3*heads*hd] accumulates the gradient reaching the bias through padded key slots.         */
int spg_attn_bwd(int dtype, const void* qkv, const void* q_pooled, const void* qkv_bias_t, const void* out,
                 const void* dout, const float* lse, void* dqkv, void* dq_pooled, float* dbias_pad, float* delta_ws,
                 int B, int H, int W, int heads, int hd, int ws, spg_stream_t stream);
/* 2x2 max pool on NHWC with channel window [c0, c0+C) of a row of ldc channels; idx u8 saved.
 * bwd scatters dy into dx (dx rows of ldc channels; only [c0,c0+C) touched; accumulate=0 overwrites).      */
int spg_maxpool2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int ldc, int c0,
                     spg_stream_t stream);
int spg_maxpool2_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, int ldc,
                     int c0, spg_stream_t stream);
/* patch embedding input gather: image f32 NCHW [B,3,H,W] (H, W % 4 == 0) -> im2col rows [B*(H/4)*(W/4), Kpad] (k = c*49+ky*7+kx,
 * zero padded to Kpad), so that patch-embed is a spg_gemm_nt; pos-embed add is the GEMM's residual.          */
int spg_patch_im2col(int dtype, const float* img, void* cols, int B, int H, int W, int Kpad, spg_stream_t stream);
/* device input pipeline (reference: CODImageProcessor.process_image, utils/image_processor.py:118-131): uint8 HWC [H,W,3] (device) ->
 * float / 255 -> antialiased bilinear resize to OH x OW (ATen _upsample_bilinear2d_aa, align_corners = false) -> (v - mean) / std,
 * written as f32 CHW [3,OH,OW].  mean3 / std3 are HOST arrays of three floats.                                                   */
int spg_preprocess_image(const uint8_t* img_hwc, float* out_chw, int H, int W, int OH, int OW, const float* mean3, const float* std3,
                         spg_stream_t stream);

/* the same for a batch of up to 64 images of different sizes in ONE launch: image i is uint8 HWC [H[i],W[i],3] at base + offs[i] (device
 * buffer `base`; offs / H / W are HOST arrays); output f32 [B,3,OH,OW] = the model's input batch (utils/data_loader.py:177-212 stacks
 * the per-image tensors the reference's DataLoader workers compute on the CPU).                                                      */
int spg_preprocess_batch(const uint8_t* base, const long* offs, const int* H, const int* W, float* out_b3hw, int B, int OH, int OW,
                         const float* mean3, const float* std3, spg_stream_t stream);

/* ---- column reductions / elementwise (HBM-bound) -----------------------------------------------------------
 * Deterministic reductions: a reduction that spans workgroups writes one partial vector per workgroup into red_ws (caller-owned
 * scratch of >= spg_reduce_workspace_floats(dtype, C, nimg) floats, contents irrelevant) and the workgroup that arrives last at a
 * counter (red_counters: >= spg_reduce_counters(dtype, C, nimg) 32-bit words that are ZERO before their first use; each launch leaves
 * them zero) adds the partials in a fixed order.  No float atomics: identical inputs give bit-identical results, which train-mode
 * BatchNorm statistics (feature_integration.py:335-345: B values per channel) need.  Launches sharing counters must be stream-ordered.
 * colsum: out[c] (+)= sum_m x[m][c] (bias gradients).  gap_sum / chan_prod_sum: out[b][c] = the same per image (overwrites).     */
long spg_reduce_workspace_floats(int dtype, int C, int nimg);
int spg_reduce_counters(int dtype, int C, int nimg);
int spg_colsum(int dtype, const void* x, float* out, int M, int C, int ldx, int accumulate, float* red_ws, long red_ws_floats,
               unsigned* red_counters, spg_stream_t stream);
int spg_gap_sum(int dtype, const void* x, float* out, int B, long HW, int C, float* red_ws, long red_ws_floats,
                unsigned* red_counters, spg_stream_t stream);
int spg_chan_prod_sum(int dtype, const void* a, const void* b, float* out, int B, long HW, int C, float* red_ws, long red_ws_floats,
                      unsigned* red_counters, spg_stream_t stream);
int spg_add(int dtype, const void* a, const void* b, void* out, long n, spg_stream_t stream);
int spg_cast_bf16(const float* f32, void* bf16, long n, int to_f32, spg_stream_t stream);
/* The cast back (bf16 -> f32, n multiple of 8) that also writes nparts (1..4096) partial sums of squares of the values it wrote: the clip's
 * norm of an all-reduced gradient range without a pass of its own (add the arrays with spg_sumsq_fold).  Deterministic per nparts. */
int spg_cast_bf16_sq(float* f32, const void* bf16, long n, float* sq_part, int nparts, spg_stream_t stream);
int spg_copy_channels(int dtype, const void* x, void* y, long M, int C, int ldx, int cx0, int ldy, int cy0,
                      int accumulate, spg_stream_t stream);

/* ---- BatchNorm2d on NHWC rows [M,C] (nn.BatchNorm2d eps 1e-5, momentum 0.1) ----------
 * bn_stats: stats[c] = sum x, stats[C+c] = sum x^2 (f32 [2C], overwritten; deterministic, see above).
 * bn_finalize: training: batch mean / biased var from stats, running stats updated with the unbiased var;
 *   eval: running stats.  Writes scale_shift f32 [2C] (y = x*scale+shift) and mean_invstd f32 [2C].
 * bn_apply: y = relu?(x*scale+shift).
 * bn_bwd_reduce: sums[c] = sum dy', sums[C+c] = sum dy'*xhat, dy' = dy where the (recomputed) ReLU passed (overwritten).
 * bn_bwd_apply: dx = gamma*invstd*(dy' - sums0/M - xhat*sums1/M); dgamma += sums1, dbeta += sums0.          */
int spg_bn_stats(int dtype, const void* x, float* stats, long M, int C, float* red_ws, long red_ws_floats, unsigned* red_counters,
                 spg_stream_t stream);
/* bn_stats + bn_finalize(training) in ONE launch: the workgroup that finishes a channel slab's sums also writes its scale_shift /
 * mean_invstd / running statistics; num_batches_tracked (int64 device scalar or NULL) += 1.  stats f32 [2C] is scratch.          */
int spg_bn_stats_finalize(int dtype, const void* x, float* stats, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, long long* num_batches_tracked, float* scale_shift, float* mean_invstd, long M, int C,
                          float eps, float momentum, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
/* ---- 3x3 convolution + BatchNorm batch statistics in one launch (bf16; the LDS-resident-input kernel of csrc/conv_halo.hip)
 * Replaces nn.Conv2d(3x3, pad 1) followed by the statistics pass of the nn.BatchNorm2d behind it (reference
 * models/object_detection.py:115-123, 193-199, 230-236).  C = conv(X) + bias as spg_gemm_nt(conv3x3=1) writes it (ldc = Co), and
 * stats_part f32 [rows][2*Co]: partial sums and sums of squares of the ROUNDED outputs over disjoint pixel sets (one row per workgroup and
 * wave row when Co is 64 or 128, per 8 x 32 pixel tile and wave row otherwise), every row written; cu_budget must match the launch's.
 * rows = spg_conv3x3_stats_rows(...) (0: no instance for the shape -- use spg_gemm_nt + spg_bn_stats_finalize).
 * spg_bn_stats_finalize_part sums the rows in a fixed order and finalises exactly as spg_bn_stats_finalize (M = B*H*W samples).      */
long spg_conv3x3_stats_rows(int dtype, int B, int H, int Wd, int Ci, int Co, int cu_budget);
int spg_conv3x3_fwd_stats(int dtype, const void* X, const void* Wp, void* C, const float* bias, float* stats_part, long part_rows,
                          int B, int H, int Wd, int Ci, int Co, int cu_budget, spg_stream_t stream);
/* Weight gradient of the same convolution on the same LDS-resident tiles: dW f32 += dY^T (*) X, in the fwd pack's layout [Co][9*Ci]
 * (torch_layout = 0) or straight into the parameter's own gradient [Co][Ci][3][3] (torch_layout = 1: no packed scratch, no unpack launch);
 * dbias f32 [Co] += column sums of dY (may be NULL).  dY NHWC [B,H,W,Co], X NHWC [B,H,W,Ci], bf16.  Deterministic (per-workgroup
 * partial blocks in the caller's workspace, summed in a fixed order by a second launch).  workspace bytes from
 * spg_conv3x3_wgrad_workspace_bytes (0: no instance -- use spg_gemm_tn(conv3x3=1)).  Replaces autograd's conv2d weight gradient.      */
long spg_conv3x3_wgrad_workspace_bytes(int dtype, int B, int H, int Wd, int Ci, int Co, int cu_budget);
int spg_conv3x3_wgrad(int dtype, const void* dY, const void* X, float* dW, float* dbias, void* workspace, long workspace_bytes,
                      int B, int H, int Wd, int Ci, int Co, int torch_layout, int cu_budget, spg_stream_t stream);
int spg_bn_stats_finalize_part(const float* part, long R, float* stats, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float* scale_shift, float* mean_invstd, long M, int C,
                               float eps, float momentum, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
/* the same for FOUR BatchNorms of C/4 channels each over one [M, C] tensor (the e-ASPP branches stored branch-major in one tensor):
 * gamma4 ... num_batches_tracked4 are HOST arrays of 4 device pointers (running_* / num_batches_tracked arrays or entries may be NULL). */
int spg_bn_stats_finalize4(int dtype, const void* x, float* stats, const float* const* gamma4, const float* const* beta4,
                           float* const* running_mean4, float* const* running_var4, long long* const* num_batches_tracked4, float* scale_shift,
                           float* mean_invstd, long M, int C, float eps, float momentum, float* red_ws, long red_ws_floats,
                           unsigned* red_counters, spg_stream_t stream);
int spg_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* scale_shift, float* mean_invstd, long M, int C, float eps,
                    float momentum, int training, spg_stream_t stream);
int spg_bn_apply(int dtype, const void* x, const float* scale_shift, void* y, long M, int C, int relu,
                 spg_stream_t stream);
int spg_bn_bwd_reduce(int dtype, const void* dy, const void* x, const float* scale_shift, const float* mean_invstd,
                      float* sums, long M, int C, int relu, float* red_ws, long red_ws_floats, unsigned* red_counters,
                      spg_stream_t stream);
int spg_bn_bwd_apply(int dtype, const void* dy, const void* x, const float* scale_shift, const float* mean_invstd,
                     const float* gamma, const float* sums, void* dx, float* dgamma, float* dbeta, long M, int C,
                     int relu, spg_stream_t stream);

/* ---- CFI (models/feature_integration.py) -----------------------------------------------------------------
 * upsample_bilinear (align_corners=False) writes into channels [c0,c0+C) of rows of ldy channels, i.e. it
 *   places its result directly inside the 2016-channel fusion input / the 320-channel PED concat
 *   (feature_integration.py:229-236, object_detection.py:219-227); _bwd is the exact adjoint (gather form).
 * se_fc: hidden = relu(W1 gap), scale = sigmoid(W2 hidden) (feature_integration.py:121-126,147-151).
 * dwconv3x3: dilated depth-wise 3x3, pad = dil (feature_integration.py:317-332); flip=1 gives the input grad.
 * easpp_fuse: grouped 1x1 over the BRANCH-MAJOR concat of 4 branches + broadcast global branch: group g reads
 *   concat channels 5g..5g+4 (feature_integration.py:348-360,411-412).                */
int spg_upsample_bilinear(int dtype, const void* x, void* y, int B, int h, int w, int C, int H, int W, int ldy,
                          int c0, spg_stream_t stream);
int spg_upsample_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int h, int w, int C, int H, int W, int ldy,
                              int c0, int accumulate, spg_stream_t stream);
/* in_scale multiplies gap on load: pass the per-image column SUMS of spg_gap_sum and 1 / HW (the squeeze is a mean,
 * feature_integration.py SE block); dgap of se_fc_bwd is the gradient w.r.t. the scaled input.                                    */
int spg_se_fc(const float* gap, const float* w1, const float* w2, float* hidden, float* scale, int B, int C, int R, float in_scale,
              spg_stream_t stream);
/* reductions across workgroups below are deterministic (partials + fixed-order finish by the last workgroup): red_ws is scratch of the
 * stated size, red_counter ONE 32-bit word that is zero before its first use (see "Deterministic reductions").
 * se_fc_bwd: B*(C+R) floats.  dwconv3x3_wgrad: 64*9*C floats.  easpp_fuse_bwd: 32*B*6*C floats (dglob is overwritten).            */
int spg_se_fc_bwd(const float* gap, const float* w1, const float* w2, const float* hidden, const float* scale,
                  const float* dscale, float* dgap, float* dw1, float* dw2, int B, int C, int R, float in_scale, float* red_ws,
                  long red_ws_floats, unsigned* red_counter, spg_stream_t stream);
/* dst[R][Kp] (dtype) = [a[R][na] | b[R][nb] | 0]: the position-embedding GEMM's weight operand [pos_embed | pos_embed_window]
 * (reference: sam2 Hiera._get_pos_embed via feature_encoding.py:236) packed in ONE launch per step.                               */
int spg_pack_cols2(int dtype, const float* a, int na, const float* b, int nb, void* dst, int R, int Kp, spg_stream_t stream);
/* up to 4 jobs of dst[r][c] += src[r][c], c < C (row strides ldd / lds): column slices of padded weight-gradient results into the
 * gradients of patch_embed.proj.weight / pos_embed / pos_embed_window.                                                            */
int spg_add_cols_batch(int njobs, float* const* dst, const float* const* src, const int* R, const int* C, const int* ldd,
                       const int* lds, spg_stream_t stream);
int spg_chan_scale(int dtype, const void* x, const float* scale, void* y, int B, long HW, int C, spg_stream_t stream);
int spg_chan_scale_bwd(int dtype, const void* dy, const float* scale, const float* dgap, void* dx, int B, long HW,
                       int C, spg_stream_t stream);
int spg_dwconv3x3(int dtype, const void* x, const float* w, void* y, int B, int H, int W, int C, int dil, int flip,
                  spg_stream_t stream);
int spg_dwconv3x3_wgrad(int dtype, const void* dy, const void* x, float* dw, int B, int H, int W, int C, int dil,
                        float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream);
int spg_easpp_fuse(int dtype, const void* br0, const void* br1, const void* br2, const void* br3, const float* glob,
                   const float* w, void* y, int B, long HW, int C, spg_stream_t stream);
int spg_easpp_fuse_bwd(int dtype, const void* dy, const void* br0, const void* br1, const void* br2, const void* br3,
                       const float* glob, const float* w, void* d0, void* d1, void* d2, void* d3, float* dglob,
                       float* dw, int B, long HW, int C, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream);

/* ---- EFE / PED heads (models/object_detection.py:126-130,155,306,339): 1x1 conv C->1 with bias ------------ */
int spg_head1x1(int dtype, const void* x, const float* w, const float* b, void* y, long M, int C, spg_stream_t stream);
long spg_head1x1_bwd_workspace_floats(int C);   /* scratch of spg_head1x1_bwd's deterministic dw / db reduction (+ one zeroed counter) */
int spg_head1x1_bwd(int dtype, const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, long M,
                    int C, int accumulate, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream);

/* ---- fused CFI / EFE / PED element kernels (csrc/head.hip) ---------------------------------------------------------------------
 * bn_apply_head: y = relu?(x*scale+shift) (y may be NULL: not stored) and pred[m] = w . y[m,:] + b[0] in one pass -- BN-apply + ReLU +
 *   the 1x1 head of object_detection.py:150-155 (EFE) and :232-236 + :339 (PED stage end).
 * ped_gather: pc[B,H,W,Cx+Ce] = cat[ up2(act(x[B,hx,wx,Cx])), up(edge[B,he,we,Ce]) ] (object_detection.py:219-232), act = relu(x*scale+shift)
 *   when x_scale_shift != NULL (the previous stage's BN-apply folded in), identity otherwise; H = 2*hx, edge scale 2 or 4; Ce may be 0.
 * ped_gather_bwd: exact adjoint of one source's bilinear upsample (scale 2 or 4), gather form, no atomics:
 *   dx[B,h,w,C] (+)= up^T(dy[B,H,W,ldy][..., c0:c0+C]).
 * bn_bwd_head: BatchNorm backward (training) with the 1x1 head's gradient formed on the fly:  dy = dnext (may be NULL) + dpred[m]*head_w[c],
 *   masked by the recomputed ReLU; dx as spg_bn_bwd_apply; dgamma, dbeta, dhead_w, dhead_b accumulate (+=).  sums: f32 [3C+1] scratch.
 *   Deterministic reduction: red_ws >= spg_bn_bwd_head_workspace_floats(dtype, C) floats, spg_bn_bwd_head_counters zeroed words.     */
/* cfi_combine: out[B,H,W,C] = y2 + up(y3[B,h3,w3,C]) + up(y4[B,h4,w4,C]) (bilinear, align_corners=False): the CFI fusion's 1x1 conv over
 *   cat[s2, up(s3), up(s4)] (feature_integration.py:229-239) evaluated as three GEMMs at each map's own resolution -- a 1x1 conv commutes
 *   with bilinear interpolation -- so the 2016-channel concat is never built.                                                          */
int spg_cfi_combine(int dtype, const void* y2, const void* y3, const void* y4, void* out, int B, int H, int W, int h3, int w3, int h4,
                    int w4, int C, spg_stream_t stream);
int spg_bn_apply_head(int dtype, const void* x, const float* scale_shift, const float* w, const float* b, void* y, void* pred, long M,
                      int C, int relu, spg_stream_t stream);
int spg_ped_gather(int dtype, const void* x, const float* x_scale_shift, int hx, int wx, int Cx, const void* edge, int he, int we, int Ce,
                   void* y, int B, int H, int W, spg_stream_t stream);
int spg_ped_gather_bwd(int dtype, const void* dy, void* dx, int B, int h, int w, int C, int H, int W, int ldy, int c0, int accumulate,
                       spg_stream_t stream);
long spg_bn_bwd_head_workspace_floats(int dtype, int C);
int spg_bn_bwd_head_counters(int dtype, int C);
int spg_bn_bwd_head(int dtype, const void* dnext, const void* x, const void* dpred, const float* head_w, const float* scale_shift,
                    const float* mean_invstd, const float* gamma, float* sums, void* dx, float* dgamma, float* dbeta, float* dhead_w,
                    float* dhead_b, long M, int C, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);

/* ---- e-ASPP middle, branch-batched (csrc/easpp.hip; feature_integration.py:397-412) -----------------------------------------------
 * The four dilated depth-wise branches share ONE tensor dcat [M, 4C] in the reference's branch-major concat order.
 * dwconv4: dcat[p][br*C+c] = dilated depth-wise 3x3 of x[B,H,W,C] with w4[br] (f32 [C][9]), dilation dil4[br] = padding (HOST arrays of 4).
 * dwconv4_dgrad: dx[p][c] = gadd[b][c] (f32 [B,C] or NULL: the global-average-pool adjoint) + sum over branches of the flipped convolution
 *   of dy[M,4C].   dwconv4_wgrad: dw4[br] (f32 [C][9]) += ...; deterministic: red_ws 4*64*9*C floats, 4 zeroed counters.
 * easpp_fuse_bn: y[p][g] = sum_j w[g][j] * cat[p][5g+j], cat = [relu(dcat*scale+shift) | glob[b]] (grouped 1x1, groups = C, branch-major
 *   quirk SURVEY 2.2 C8) with the four branch BatchNorms' apply + ReLU folded in (scale_shift f32 [2*4C]).
 * easpp_fuse_bn_bwd: gradient of that w.r.t. dcat THROUGH the branch BatchNorms (training statistics) in one reduce + apply pair:
 *   ddcat[M,4C]; dgamma4/dbeta4 (four f32 [C], +=), dw[cc] += for cc < 4C (the global-branch part of dw and dglob: easpp_global_bwd).
 * easpp_global_fwd/bwd: the global branch (GAP -> 1x1 conv C->C -> BatchNorm over the B values -> ReLU, :335-345,401-408) as one
 *   single-workgroup kernel each way; gsum = per-image column sums of the reduced map (spg_gap_sum), S = those of the fusion conv's
 *   output gradient; gadd = d(GAP input)/HW for dwconv4_dgrad.  B <= 64.                                                              */
int spg_dwconv4(int dtype, const void* x, const float* const* w4, const int* dil4, void* dcat, int B, int H, int W, int C, spg_stream_t stream);
int spg_dwconv4_dgrad(int dtype, const void* dy, const float* const* w4, const int* dil4, const float* gadd, void* dx, int B, int H, int W, int C,
                      spg_stream_t stream);
int spg_dwconv4_wgrad(int dtype, const void* dy, const void* x, const int* dil4, float* const* dw4, int B, int H, int W, int C, float* red_ws,
                      long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
int spg_easpp_fuse_bn(int dtype, const void* dcat, const float* scale_shift, const float* glob, const float* w, void* y, int B, long HW, int C,
                      spg_stream_t stream);
long spg_easpp_fuse_bn_bwd_workspace_floats(int dtype, int C);
int spg_easpp_fuse_bn_bwd_counters(int dtype, int C);
int spg_easpp_fuse_bn_bwd(int dtype, const void* dfu, const void* dcat, const float* w, const float* scale_shift, const float* mean_invstd,
                          const float* const* gamma4, float* const* dgamma4, float* const* dbeta4, float* sums, void* ddcat, float* dw, int B,
                          long HW, int C, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
int spg_easpp_global_fwd(const float* gsum, const float* Wg, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* gm, float* gl0, float* glob, float* scale_shift, float* mean_invstd, int B, int C,
                         long HW, float eps, float momentum, int training, spg_stream_t stream);
int spg_easpp_global_bwd(const float* S, const float* glob, const float* gl0, const float* gm, const float* wf, const float* Wg,
                         const float* gamma, const float* mean_invstd, float* dwf, float* dWg, float* dgamma, float* dbeta, float* gadd, int B,
                         int C, long HW, int training, spg_stream_t stream);

/* ---- CODLoss, fixed-size ground truth (utils/loss_functions.py:114-295 + resize loop engine/trainer.py:358-383) ----------
 * weight_map: w = 1 + bw*(|Laplace3x3 m| + |avgpool31 m - m|); stats[b] = {sum m, sum w, sum edge_gt, -} (overwritten).
 * loss_reduce: per image sums at the TARGET resolution of the bilinearly resized logits: edge=0 -> {sum w*bce, sum s*m*w,
 *   sum (s+m)*w}; edge=1 -> {sum focal, sum s*t, sum s} (sums f32 [B][3], overwritten).  Both reductions are deterministic
 *   (partials + fixed-order finish): red_ws >= spg_loss_workspace_floats(B, S) floats, red_counters = B zeroed words (see above).
 * loss_finalize: out = {loss, seg_loss, edge_loss} from stats + seg_sums[3][B][3] + edge_sums[B][3].
 * loss_grad: d loss / d prediction at the prediction's own resolution (bilinear adjoint in gather form), times
 *   coef * grad_out[0] (grad_out may be NULL = 1).                                                                    */
long spg_loss_workspace_floats(int B, int S);
int spg_loss_weight_map(const float* mask, const float* edge_gt, float* wmap, float* stats, int B, int S,
                        float boundary_weight, float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
int spg_loss_reduce(int dtype, const void* pred, const float* target, const float* wmap, const float* stats,
                    float* sums, int B, int S, int h, int w, int edge, float alpha, float gamma, float* red_ws,
                    long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
int spg_loss_finalize(const float* stats, const float* seg_sums, const float* edge_sums, float* out, int B, int S,
                      float sw0, float sw1, float sw2, float bce_w, float iou_w, float edge_w, spg_stream_t stream);
/* dz_ws: B*S*S floats of scratch or NULL.  With it (and a prediction coarser than the target) the gradient runs as two passes -- the
 * loss derivative of every full-res pixel once, then the adjoint of the bilinear up-sampling -- instead of re-evaluating the
 * derivative for every low-res logit whose window covers a pixel (9x fewer evaluations at the trunk's three scales).              */
int spg_loss_grad(int dtype, const void* pred, const float* target, const float* wmap, const float* stats,
                  const float* sums, const float* grad_out, void* dpred, int B, int S, int h, int w, int edge, float coef,
                  float bce_w, float iou_w, float alpha, float gamma, float* dz_ws, spg_stream_t stream);

/* The same reductions / gradients for the loss's four maps in one launch each (preds, hs, ws, dpreds, coefs: HOST arrays of 4 --
 * [0..2] the segmentation logits against `masks` (seg_sums[3][B][3]), [3] the edge logits against `edge_gt` (edge_sums[B][3]); read
 * at the call).  Results are bit-identical to four spg_loss_reduce / spg_loss_grad calls (same blocks per map, same fixed-order finish).
 * reduce_all: red_ws >= spg_loss_reduce_all_workspace_floats(B) floats, red_counters = 4 B zeroed words.
 * grad_all: dz_ws = 4 * B * S * S floats; coefs[i] = scale weight / B (i < 3), edge weight / B (i = 3); two launches (the derivative per
 *   full-res pixel of every map -- which IS the gradient of a map already at S x S --, then the bilinear adjoint of the coarser maps). */
long spg_loss_reduce_all_workspace_floats(int B);
int spg_loss_reduce_all(int dtype, const void* const* preds, const int* hs, const int* ws, const float* masks, const float* edge_gt,
                        const float* wmap, const float* stats, float* seg_sums, float* edge_sums, int B, int S, float alpha, float gamma,
                        float* red_ws, long red_ws_floats, unsigned* red_counters, spg_stream_t stream);
int spg_loss_grad_all(int dtype, const void* const* preds, void* const* dpreds, const int* hs, const int* ws, const float* coefs,
                      const float* masks, const float* edge_gt, const float* wmap, const float* stats, const float* seg_sums,
                      const float* edge_sums, const float* grad_out, int B, int S, float bce_w, float iou_w, float alpha, float gamma,
                      float* dz_ws, spg_stream_t stream);

/* Warm-up hint for the launch AFTER the next spg_gemm_nt: `bytes` at `next` (typically the weight matrix of the following Linear, which
 * the step reads from HBM exactly once per pass) are requested -- one load per 128-byte line, never waited for -- by the workgroups of the
 * spg_gemm_nt call that follows this one on the same host thread, so that they sit in the Infinity Cache when the following launch
 * starts.  No effect on results; consumed by exactly one spg_gemm_nt call (the bf16 dense / implicit-GEMM kernels honour it, the other
 * families ignore it); NULL / 0 clears it.  (models/feature_encoding.py:236: the order of a MultiScaleBlock's Linear layers is static.) */
int spg_prefetch_hint(const void* next, long bytes);

/* ---- optimizer (engine/trainer.py:274-306 param groups, :399-409 clip + AdamW step) over a flat f32 arena ---------
 * sumsq: out[0] = sum x^2 (deterministic: 2048 floats of scratch + one zeroed counter, see "Deterministic reductions").  adamw: step_f[0] += 1, then clip coefficient min(1, clip/(sqrt(gnorm_sq)*grad_scale+1e-6))
 * and a decoupled-weight-decay Adam update; group_of_chunk[i/256] selects lr[g], wd[g] (device arrays, so the
 * scheduler can change them without re-capturing a hipGraph).  Every parameter starts on a 256-element boundary.
 * zero_grad != 0 clears g after use (the next step's kernels accumulate into it), saving a separate memset pass.      */
int spg_sumsq(const float* x, float* out, long n, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream);
/* The same norm when spg_gemm_tn_blocks has already produced the sums of squares of the gradients it wrote: out[0] = sum of squares over
 * `chunks` -- a DEVICE array of nchunks records struct { long off; int n4; int pad; } (16 bytes): n4 <= 4096 groups of four floats at
 * x + off, disjoint, together the gradients NOT covered by the extras -- plus the sums of the nextras (<= 32) arrays extra_ptr[i][0 .. extra_n[i])
 * (HOST arrays of device pointers / lengths: the sq_part outputs of the step's spg_gemm_tn_blocks launches).  Fixed summation order:
 * deterministic.  red_ws: >= nchunks floats.                                                                                          */
int spg_sumsq_fold(const float* x, const void* chunks, int nchunks, int nextras, const float* const* extra_ptr, const int* extra_n,
                   float* out, float* red_ws, long red_ws_floats, unsigned* red_counter, spg_stream_t stream);
int spg_adamw(float* p, float* g, float* m, float* v, const unsigned char* group_of_chunk, const float* lr,
              const float* wd, const float* gnorm_sq, float* step_f, float clip, float beta1, float beta2, float eps,
              float grad_scale, int zero_grad, long n, spg_stream_t stream);

/* adamw fused with the per-step weight re-pack (replaces spg_adamw + spg_pack_batch + spg_pack_conv3x3 inside a train step): while
 * the new parameter values are in registers the kernel also writes the compute-dtype copies the GEMMs read.  jobs = DEVICE array of
 *   struct { long off; void* dst; void* dst_t; int R, C, lds, ldd, item0, kind; }          (48 bytes)
 * kind 1 (matrix): source element (r, c) at arena offset off + r*lds + c; dst[r*ldd + c] and dst_t[c*R + r] (either may be NULL);
 *   work items = 64 x 64 tiles.  kind 0 (flat, R = 1, C = n elements): dst[i] copy or NULL; kind 2 (3x3 convolution [R = Co][C = Ci][3][3]):
 *   dst = [Co][tap][Ci], dst_t = [Ci][8 - tap][Co]; both in 4096-element work items.  item0 = prefix sum of work items, total_items their
 *   sum; the jobs must cover every parameter exactly once.  Other arguments as spg_adamw.                                           */
int spg_adamw_pack(int dtype, float* p, float* g, float* m, float* v, const unsigned char* group_of_chunk, const float* lr,
                   const float* wd, const float* gnorm_sq, float* step_f, float clip, float beta1, float beta2, float eps,
                   float grad_scale, int zero_grad, const void* jobs, int njobs, int total_items, spg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
