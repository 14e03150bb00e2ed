"""Forward/backward orchestration of the SPEGNet hot path over the HIP kernels (spegnet_amd.ops).

Everything numeric here is a call into libspegnet_hip.so; Python only sequences launches and owns the saved
activations.  Layout: NHWC activations in the compute dtype T (float32 = parity mode, bfloat16 = fast mode),
fp32 master parameters, T copies of the matrix weights re-packed once per optimizer step (`pack()`).

Reference anchors: SPEGNet.forward models/spegnet.py:137-206; Hiera trunk via models/feature_encoding.py:236
(algorithm: SURVEY.md §8 row E); CFI models/feature_integration.py:128-151,205-246,369-417; EFE/PED
models/object_detection.py:132-157,201-238,309-342.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os
import torch
import torch.nn.functional as F

from .. import ops
from .params import block_table

Tensor = torch.Tensor
EASPP_RATES = (1, 6, 12, 18)
FUSION_W = "fusion.conv1x1.weight"
PATCH_KPAD = 160  # 3*7*7 = 147 padded to a multiple of 16 elements


class BNState:
    """Per-call BatchNorm record (train: batch statistics; eval: running statistics)."""
    __slots__ = ("x", "ss", "mi", "C", "relu", "prefix")


class Engine:
    def __init__(self, module, cfg, dtype: torch.dtype):
        self.m = module
        self.cfg = cfg
        self.dtype = dtype
        self.blocks = block_table(cfg)
        self.P: Dict[str, Tensor] = {}
        self.W: Dict[str, Tensor] = {}      # packed T copies
        self.unit_cb = None   # called with the index of each finished backward unit (head=0, then blocks last-to-first)
        # Weight-gradient GEMMs feed nothing but the optimizer, and the persistent GEMM kernels rarely fill all 256 CUs (tile / split
        # quantisation, ramp-up and tail of each launch).  With wgrad_async they are forked onto a second HIP stream -- inside a
        # captured step that is a parallel branch of the graph -- so their workgroups could pack into CUs the dgrad chain leaves idle.
        # Operands are held until join_wgrad(), which every backward piece calls before it returns.  Off by default: on one MI355X
        # the two chip-sized branches contend (226.0 vs 232.2 img/s, see engine/trainer.py).  A SPATIAL split was measured too
        # (round 2: the grouped wgrad launches on a side stream with a 64 / 96-CU grid, the dgrad chain sized for the other CUs):
        # 34.3 / 31.9 ms per step against 29.2 in line -- the wgrad branch becomes the critical path before the chain's idle CU time
        # pays for the partition.
        self.wgrad_async = False
        self.group_wgrads = True    # trunk blocks: one grouped wgrad launch per block (False: per-layer launches, for A/B runs)
        self._wg_jobs = None
        # Whole-block wgrads (ops.gemm_tn_blocks): the weight gradients of CONSECUTIVE trunk blocks are held back until their 256 x 192
        # blocks of dW just fill the CUs (Hiera-L stage 3: 84 per trunk block, three trunk blocks = 252 for 256 CUs) and go out as one
        # launch in which every workgroup owns a whole block over all of M -- no partial sums, no reduce.  Their dY / X operands stay
        # referenced by the pending list until then.  Not used while a per-unit gradient callback needs finished ranges mid-backward.
        self.block_wgrads = os.environ.get("SPG_BLOCK_WGRADS", "1") != "0"
        self._wg_pending = {}       # M -> [(jobs of one trunk block with that row count, their block count)]
        # Inside a single-GPU TrainStep (gradients are zero when the backward starts, the clip's norm is taken right after it): the
        # whole-block launches STORE their blocks instead of adding (no cold read of the old values at the exit) and leave the sum of
        # squares of what they wrote, so that the optimizer's global-norm pass reads only the gradients no such launch covered
        # (Arena.step(fold=...), spg_sumsq_fold).  Set and cleared by TrainStep around each step; never on for plain autograd use.
        self.fold_sumsq = False      # whole-block wgrad launches also hand back their sums of squares (single GPU: the clip's norm pass skips them)
        self.store_wgrads = False    # ... and STORE their blocks instead of adding to them (the caller keeps those gradients uncleared: Arena.step(keep=))
        self._sq_parts, self._sq_cover = [], []
        self.batch_ln_params = True  # trunk: LayerNorm dgamma / dbeta in batched launches
        self._ln_jobs = []
        self._tn_defer = []
        self._side = None
        self._held = []
        self._forked = False
        self.refresh_params()

    # ------------------------------------------------------------------------------------------------
    def refresh_params(self):
        self.P = dict(self.m.named_parameters())
        self.P.update(dict(self.m.named_buffers()))

    def grad(self, name: str) -> Tensor:
        p = self.P[name]
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def pack(self):
        """fp32 master -> compute-dtype copies ([N,K] and [K,N]; 3x3 convs in [Co][tap][Ci] / dgrad form).  All plain
        matrices go through ONE batched launch (job table built once; parameter storage is stable)."""
        T, P, W = self.dtype, self.P, self.W
        if getattr(self, "_pack_jobs", None) is None or self._pack_key != tuple(p.data_ptr() for p in P.values()):
            import struct
            jobs, tile0 = [], 0
            self._pack_conv, self._pack_keep = [], []
            for name, p in P.items():
                if not name.endswith(".weight") or p.dim() < 2:
                    continue
                if name.endswith("patch_embed.proj.weight") or name == FUSION_W:
                    continue
                if p.dim() == 4 and p.shape[2] == 3 and p.shape[1] > 1:
                    self._pack_conv.append(name)
                    continue
                if (p.dim() == 4 and p.shape[1] == 1) or any(k in name for k in ("se_block", "global_branch", "context.fusion.0", "pred_heads", "edge_conv")):
                    continue  # depth-wise / tiny layers are read as fp32 by their kernels
                R, C = p.shape[0], p.numel() // p.shape[0]
                W[name] = torch.empty((R, C), dtype=T, device=p.device)
                W[name + ":T"] = torch.empty((C, R), dtype=T, device=p.device)
                jobs.append(struct.pack("<QQQiiii", p.data_ptr(), W[name].data_ptr(), W[name + ":T"].data_ptr(), R, C, tile0, 0))
                tile0 += ((R + 31) // 32) * ((C + 31) // 32)
            # the per-block qkv biases the attention kernels read in the compute dtype ride in the same launch (1 x C "matrices")
            for b in self.blocks:
                n = f"encoder.encoder.blocks.{b['idx']}.attn.qkv.bias"
                p = P[n]
                W[n] = torch.empty(p.shape, dtype=T, device=p.device)
                jobs.append(struct.pack("<QQQiiii", p.data_ptr(), W[n].data_ptr(), 0, 1, p.numel(), tile0, 0))
                tile0 += (p.numel() + 31) // 32
            blob = torch.frombuffer(bytearray(b"".join(jobs)), dtype=torch.uint8).to(next(iter(P.values())).device)
            self._pack_jobs, self._pack_n, self._pack_tiles = blob, len(jobs), tile0
            self._pack_key = tuple(p.data_ptr() for p in P.values())
        ops.pack_batch(self._pack_jobs, self._pack_n, self._pack_tiles, T)
        for name in self._pack_conv:
            f, d = ops.pack_conv3x3(P[name].detach(), T, W.get(name), W.get(name + ":dgrad"))
            W[name], W[name + ":dgrad"] = f, d
        # CFI fusion conv, split by input map ([512, 288 | 576 | 1152]): one [N,K] / [K,N] pair per source (head_fwd evaluates the conv per
        # source resolution); per step these copies are written by the fused optimizer (opt_jobs), this path runs at load time only
        fw = P[FUSION_W].detach().view(P[FUSION_W].shape[0], -1)
        for tag, (a, b) in self.fusion_split().items():
            sub = fw[:, a:b].contiguous()
            W[FUSION_W + tag] = ops.pack_matrix(sub, T, out=W.get(FUSION_W + tag))
            W[FUSION_W + tag + "T"] = ops.pack_matrix(sub, T, transpose=True, out=W.get(FUSION_W + tag + "T"))
        e = "encoder.encoder.patch_embed.proj.weight"
        pw = P[e]
        w2 = torch.zeros((pw.shape[0], PATCH_KPAD), dtype=torch.float32, device=pw.device)
        w2[:, :147] = pw.detach().reshape(pw.shape[0], 147)
        W[e] = ops.pack_matrix(w2, T, out=W.get(e))

    def fusion_split(self):
        """column ranges of fusion.conv1x1.weight [512, C2+C3+C4] belonging to s2 / s3 / s4"""
        d = self.cfg["embed_dim"]
        c2, c3, c4 = 2 * d, 4 * d, 8 * d
        return {":s2": (0, c2), ":s3": (c2, c2 + c3), ":s4": (c2 + c3, c2 + c3 + c4)}

    def opt_jobs(self, arena, keep_offsets=frozenset()):
        """Job table of spg_adamw_pack (AdamW fused with this re-pack): (device blob, jobs, work items).  Covers every parameter of
        the arena once: matrices as 64 x 64 tiles with their [N,K] / [K,N] copies, 3x3 convolutions and everything else as flat
        4096-element chunks (runs of parameters that need no copy are merged into one flat job: the arena is contiguous)."""
        import struct
        key = (arena.p.data_ptr(), tuple(t.data_ptr() for t in self.W.values()), keep_offsets)
        if getattr(self, "_opt_jobs_key", None) == key:
            return self._opt_jobs
        # keep_offsets: arena offsets of the matrices whose gradients the optimizer kernel must NOT clear (kind | 256): the next backward's
        # whole-block launches store them whole (Arena.step(fold=...), TrainStep)
        if not self.W:
            self.pack()
        P, W = self.P, self.W
        recs, item0 = [], 0
        run = None     # [off, end] of a run of copy-less parameters

        def flush():
            nonlocal run, item0
            if run is not None:
                n = run[1] - run[0]
                recs.append(struct.pack("<qQQiiiiii", run[0], 0, 0, 1, n, n, n, item0, 0))
                item0 += (n + 4095) // 4096
                run = None

        e = "encoder.encoder.patch_embed.proj.weight"
        for name in arena.names:
            p, off = P[name], arena.offsets[name]
            n = p.numel()
            rec = None
            if name == FUSION_W:       # three column slices of one parameter (source row stride = its full width)
                flush()
                R, Ct = p.shape[0], n // p.shape[0]
                for tag, (a, b) in self.fusion_split().items():
                    recs.append(struct.pack("<qQQiiiiii", off + a, W[name + tag].data_ptr(), W[name + tag + "T"].data_ptr(), R, b - a, Ct, b - a, item0, 1))
                    item0 += ((R + 63) // 64) * ((b - a + 63) // 64)
                continue
            if name == e:
                rec = (off, W[e].data_ptr(), 0, p.shape[0], 147, 147, PATCH_KPAD, 1)
            elif name + ":dgrad" in W:
                rec = (off, W[name].data_ptr(), W[name + ":dgrad"].data_ptr(), p.shape[0], p.shape[1], 0, 0, 2)
            elif name + ":T" in W:
                R = p.shape[0]
                C = n // R
                rec = (off, W[name].data_ptr(), W[name + ":T"].data_ptr(), R, C, C, C, 1 | (256 if off in keep_offsets else 0))
            elif name in W:      # qkv bias: a compute-dtype copy of a vector
                rec = (off, W[name].data_ptr(), 0, 1, n, n, n, 0)
            if rec is None:
                if run is not None and run[1] <= off:
                    run[1] = off + n
                else:
                    flush()
                    run = [off, off + n]
                continue
            flush()
            o, d, dt_, R, C, lds, ldd, kind = rec
            recs.append(struct.pack("<qQQiiiiii", o, d, dt_, R, C, lds, ldd, item0, kind))
            if kind & 255 == 1:
                item0 += ((R + 63) // 64) * ((C + 63) // 64)
            elif kind == 2:
                item0 += (R * C * 9 + 4095) // 4096
            else:
                item0 += (C + 4095) // 4096
        flush()
        blob = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(arena.p.device)
        self._opt_jobs_key, self._opt_jobs = key, (blob, len(recs), item0)
        return self._opt_jobs

    # ------------------------------------------------------------------------------------------------ BN
    def _bn_train_stats(self, prefix: str, x: Tensor, C: int, part: Optional[Tensor]):
        """batch statistics -> (scale_shift, mean_invstd) + running statistics: from the producing convolution's partial rows when it
        wrote them (no pass over x), else by the reduction over x"""
        P = self.P
        args = (P[prefix + "weight"], P[prefix + "bias"], P[prefix + "running_mean"], P[prefix + "running_var"], P[prefix + "num_batches_tracked"])
        if part is not None:
            return ops.bn_stats_finalize_part(part, x.numel() // C, C, *args)
        return ops.bn_stats_finalize(x, C, *args)

    def bn_fwd(self, prefix: str, x: Tensor, C: int, relu: bool, training: bool, save: bool, part: Optional[Tensor] = None):
        P = self.P
        M = x.numel() // C
        if training:
            ss, mi = self._bn_train_stats(prefix, x, C, part)
        else:
            ss, mi = ops.bn_finalize(None, P[prefix + "weight"], P[prefix + "bias"], P[prefix + "running_mean"],
                                     P[prefix + "running_var"], M, False)
        y = ops.bn_apply(x, ss, C, relu)
        st = None
        if save:
            st = BNState()
            st.x, st.ss, st.mi, st.C, st.relu, st.prefix = x, ss, mi, C, relu, prefix
        return y, st

    def bn_prepare(self, prefix: str, x: Tensor, C: int, relu: bool, training: bool, save: bool, part: Optional[Tensor] = None) -> BNState:
        """Statistics / scale-shift only: the apply is fused into whichever kernel consumes x (bn_apply_head, ped_gather)."""
        P = self.P
        M = x.numel() // C
        if training:
            ss, mi = self._bn_train_stats(prefix, x, C, part)
        else:
            ss, mi = ops.bn_finalize(None, P[prefix + "weight"], P[prefix + "bias"], P[prefix + "running_mean"],
                                     P[prefix + "running_var"], M, False)
        st = BNState()
        st.x, st.ss, st.mi, st.C, st.relu, st.prefix = (x if save else None), ss, mi, C, relu, prefix
        return st

    def bn_bwd(self, st: BNState, dy: Tensor) -> Tensor:
        return ops.bn_bwd(dy, st.x, st.ss, st.mi, self.P[st.prefix + "weight"], self.grad(st.prefix + "weight"),
                          self.grad(st.prefix + "bias"), st.C, st.relu)

    # ------------------------------------------------------------------------------------------------ forked weight gradients
    def _wgrad(self, launch, *keep) -> None:
        if not self.wgrad_async or self.unit_cb is not None:
            launch()
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())   # the producers of dy / x
        with torch.cuda.stream(self._side):
            launch()
        self._held.extend(keep)
        self._forked = True

    def join_wgrad(self) -> None:
        if self._forked:
            torch.cuda.current_stream().wait_stream(self._side)
            self._held.clear()
            self._forked = False

    # ------------------------------------------------------------------------------------------------ LayerNorm backward
    def ln_bwd(self, dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None):
        """dx now; the parameter gradients (which only feed the optimizer) are collected and issued as batched launches by
        flush_ln_params() -- unless a per-unit gradient callback needs every finished range complete mid-backward."""
        if not self.batch_ln_params or self.unit_cb is not None:
            return ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres)
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, dres=dres)
        self._ln_jobs.append((dy, x, mean, rstd, dgamma, dbeta))
        return dx

    def flush_ln_params(self) -> None:
        """Issues everything the trunk backward deferred because only the optimizer needs it: LayerNorm parameter gradients and
        the slab reduces of the grouped weight gradients."""
        if self._ln_jobs:
            jobs, self._ln_jobs = self._ln_jobs, []
            ops.layernorm_param_grads_batch(jobs)
        self.flush_block_wgrads()
        if self._tn_defer:
            ops.gemm_tn_group_reduce(self._tn_defer)

    # ------------------------------------------------------------------------------------------------ deferred whole-block wgrads
    def _issue_block_wgrads(self, jobs, defer_ok: bool = True) -> None:
        ops.gemm_tn_group(jobs, self._tn_defer if (self.unit_cb is None and defer_ok) else None)

    @staticmethod
    def _round_fill(total: int, cus: int) -> float:
        """Share of the (workgroup, round) slots that own a block when `total` blocks run on `cus` CUs."""
        return total / float(-(-total // cus) * cus)

    @staticmethod
    def _wg_cus():
        """(CUs a whole-block launch may fill, the round fill a pending set must reach to go out as one)"""
        budget = ops.cu_budget_now()
        return (budget, 0.65) if budget else (ops.num_cus(), 0.75)

    def queue_block_wgrads(self, jobs) -> None:
        """One trunk block's wgrad jobs.  Problems outside the whole-block kernel's domain go out now (grouped tile kernel); the others
        join the pending set of their row count M and wait until the set fills whole rounds of CUs.  (A transition block has problems
        with two different M: each part joins its own set -- block 44's M = 4608 part rides with the first stage-3 blocks.)"""
        use = self.block_wgrads and self.unit_cb is None and not self.wgrad_async
        # Under a CU budget (graph segments that replay beside a collective: 240 of 256 CUs) a stage-3 trunk block's 84 blocks fill whole
        # rounds worse -- two trunk blocks are 168 of 240 (0.70), three would spill 12 blocks into a second round -- but falling back to the
        # grouped tile kernel for all of them cost the N > 1 step 1.3 ms (one-rank RCCL rehearsal: 3.08 ms of gemm_tn_group4 in 44 launches
        # where the single-GPU step has 0.73 in 11, profiles/round4_multi_gpu_rehearsal.md): the fill a set must reach is lowered instead ...
        by_m, now = {}, []
        for j in jobs:
            M = j[0].shape[0]
            if use and M >= 1024 and ops.tn_blocks_count([j]) > 0:
                by_m.setdefault(M, []).append(j)
            else:
                now.append(j)
        if now:
            self._issue_block_wgrads(now)
        cus, need = self._wg_cus()
        if ops.cu_budget_now():
            # ... and the sets are packed matrix by matrix: a launch takes whole matrices until the next one would open a second round of
            # the budgeted CUs (stage 3: two trunk blocks + fc2, fc1, proj of the third = 231 of 240; the third's qkv starts the next set)
            for M, part in by_m.items():
                for j in part:
                    c = ops.tn_blocks_count([j])
                    pend = self._wg_pending.setdefault(M, [])
                    total = sum(n for _, n in pend)
                    if pend and (len(pend) + 1 > ops.TN_BLOCKS_MAX or (total + c > cus and self._round_fill(total, cus) >= need)):
                        self._flush_pending(M)
                        pend = self._wg_pending.setdefault(M, [])
                    pend.append(([j], c))
            return
        for M, part in by_m.items():
            cnt = ops.tn_blocks_count(part)
            pend = self._wg_pending.setdefault(M, [])
            total = sum(c for _, c in pend)
            if pend and (sum(len(j) for j, _ in pend) + len(part) > ops.TN_BLOCKS_MAX or
                         (self._round_fill(total + cnt, cus) < self._round_fill(total, cus) and self._round_fill(total, cus) >= need)):
                self._flush_pending(M)       # the set is as full as it gets: another trunk block would open a nearly empty round
                pend = self._wg_pending.setdefault(M, [])
            pend.append((part, cnt))
            if self._round_fill(sum(c for _, c in pend), cus) >= 0.9:              # the rounds are full: go now, release the operands
                self._flush_pending(M)

    def _flush_pending(self, M) -> None:
        pend = self._wg_pending.pop(M, [])
        if not pend:
            return
        cus, need = self._wg_cus()
        if self._round_fill(sum(c for _, c in pend), cus) >= need:
            jobs = [j for jobs, _ in pend for j in jobs]
            if self.fold_sumsq or self.store_wgrads:
                parts = ops.gemm_tn_blocks(jobs, overwrite=True, want_sq=self.fold_sumsq)
                if self.fold_sumsq:
                    self._sq_parts.append(parts)
                self._sq_cover.extend(t for j in jobs for t in (j[2], j[3]) if t is not None)
            else:
                ops.gemm_tn_blocks(jobs)
        elif ops.cu_budget_now():       # (single matrices pending: one grouped tile launch for all of them)
            self._issue_block_wgrads([j for jobs, _ in pend for j in jobs])
        else:
            for jobs, _ in pend:
                self._issue_block_wgrads(jobs)

    def take_sq(self):
        """(per-launch sums of squares, the gradient tensors they cover) of this backward's whole-block launches; empties the record."""
        r = (self._sq_parts, self._sq_cover)
        self._sq_parts, self._sq_cover = [], []
        return r

    def flush_block_wgrads(self) -> None:
        for M in list(self._wg_pending.keys()):
            self._flush_pending(M)

    # ------------------------------------------------------------------------------------------------ linear helpers
    def lin_bwd(self, name: str, dy: Tensor, x: Tensor, need_dx: bool = True, gelu_h: Optional[Tensor] = None,
                residual: Optional[Tensor] = None, bias: bool = True, h_is_grad: bool = False,
                prefetch: Optional[str] = None) -> Optional[Tensor]:
        """dW += dy^T x, db += colsum(dy), returns dx = dy W (optionally * gelu'(h), + residual)."""
        gw, gb = self.grad(name + ".weight").view(dy.shape[-1], -1), (self.grad(name + ".bias") if bias else None)
        if self._wg_jobs is not None:
            # inside a trunk block: the weight gradient is only collected here; block_bwd issues the block's four wgrads as ONE
            # grouped, CU-balanced launch (ops.gemm_tn_group) once the last of them is known
            self._wg_jobs.append((dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]), gw, gb))
        else:
            self._wgrad(lambda: ops.gemm_tn(dy, x, gw, dbias=gb), dy, x)
        if not need_dx:
            return None
        act = ops.ACT_MUL_H if (gelu_h is not None and h_is_grad) else ops.ACT_NONE    # gelu_h: saved derivative | pre-activation
        # prefetch: name of the Linear whose dgrad GEMM comes next -- this launch warms that layer's transposed weight copy
        return ops.gemm_nt(dy, self.W[name + ".weight:T"], gelu_h=gelu_h, residual=residual, act=act,
                           prefetch=self.W.get(prefetch + ".weight:T") if prefetch else None)

    # ================================================================================================ trunk
    def pos_basis(self, h: int, w: int, B: int) -> Tensor:
        """The position embedding bicubic(pos_embed -> h x w) + tile(pos_embed_window) is LINEAR in its two parameters:
        pos[hw, C] = Abasis[hw, 49 + 64] . [pos_embed | pos_embed_window]^T.  Abasis depends only on the sizes, so it
        is built once (by interpolating an identity) and both the embedding and its gradient become one small GEMM
        on the HIP path instead of a bicubic kernel and its atomic-scatter backward."""
        key = (h, w, B)
        if getattr(self, "_basis_key", None) == key:
            return self._basis
        bh, bw = self.cfg["bkg"]
        w0 = self.cfg["window_spec"][0]
        dev = self.P["fusion.bn.weight"].device
        n_b, n_w = bh * bw, w0 * w0
        with torch.no_grad():
            eye = torch.eye(n_b, device=dev).view(n_b, 1, bh, bw)
            A = F.interpolate(eye, size=(h, w), mode="bicubic").view(n_b, h * w).t()          # [hw, 49]
            yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
            Tm = F.one_hot(((yy % w0) * w0 + (xx % w0)).reshape(-1), n_w).float()               # [hw, 64]
            K = n_b + n_w
            Kp = (K + 7) // 8 * 8
            Ab = torch.zeros((h * w, Kp), device=dev)
            Ab[:, :n_b], Ab[:, n_b:K] = A, Tm
            self._basis = Ab.to(self.dtype).repeat(B, 1).contiguous()                           # [B*hw, Kp]
        self._basis_key, self._basis_dims = key, (n_b, n_w, Kp)
        return self._basis

    def pos_params_packed(self) -> Tensor:
        e = "encoder.encoder."
        n_b, n_w, Kp = self._basis_dims
        C = self.cfg["embed_dim"]
        return ops.pack_cols2(self.P[e + "pos_embed"].detach().view(C, n_b), self.P[e + "pos_embed_window"].detach().view(C, n_w), Kp, self.dtype)

    def trunk_fwd(self, img: Tensor, training: bool, save: bool):
        e = "encoder.encoder."
        P, W, T = self.P, self.W, self.dtype
        B, _, Hi, Wi = img.shape
        h, w = Hi // 4, Wi // 4
        D = self.cfg["embed_dim"]
        cols = ops.patch_im2col(img.float().contiguous(), T, PATCH_KPAD)
        basis = self.pos_basis(h, w, B)
        posT = ops.gemm_nt(basis, self.pos_params_packed())                       # [B*h*w, D]
        x = ops.gemm_nt(cols, W[e + "patch_embed.proj.weight"], bias=P[e + "patch_embed.proj.bias"], residual=posT)
        x = x.view(B, h, w, D)
        ctx = {"cols": cols, "basis": basis, "blocks": [], "B": B}
        feats = []
        H, Wd = h, w
        for b in self.blocks:
            x, c = self.block_fwd(b, x, B, H, Wd, save)
            if b["q_stride"]:
                H, Wd = H // 2, Wd // 2
            ctx["blocks"].append(c)
            if b["stage_end"]:
                feats.append(x)
        return feats, ctx

    def block_fwd(self, b, x: Tensor, B: int, H: int, Wd: int, save: bool):
        p = f"encoder.encoder.blocks.{b['idx']}."
        P, W = self.P, self.W
        dim, do, heads, ws, qs = b["dim"], b["dim_out"], b["heads"], b["window"], b["q_stride"]
        hd = do // heads
        eps = self.cfg["ln_eps"]
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, P[p + "norm1.weight"], P[p + "norm1.bias"], eps)
        sc_idx = None
        if dim != do:
            sc_full = ops.gemm_nt(ln1, W[p + "proj.weight"], bias=P[p + "proj.bias"])
            shortcut, sc_idx = ops.maxpool2_fwd(sc_full, B, H, Wd, do, do, 0)
        else:
            shortcut = x
        # (every GEMM of the block warms the weight matrix of the GEMM after it: the step reads each bf16 weight copy from HBM exactly once
        # per pass, at the head of a launch whose other operand is warm -- ops.gemm_nt(prefetch=...), csrc/gemm.hip: prefetch_lines)
        qkv = ops.gemm_nt(ln1, W[p + "attn.qkv.weight"], bias=P[p + "attn.qkv.bias"], prefetch=W[p + "attn.proj.weight"])
        qp = q_idx = None
        if qs:
            qp, q_idx = ops.maxpool2_fwd(qkv, B, H, Wd, do, 3 * do, 0)
        att, lse = ops.attn_fwd(qkv, W[p + "attn.qkv.bias"], B, H, Wd, heads, hd, ws, q_pooled=qp)
        Hq, Wq = (H // 2, Wd // 2) if qs else (H, Wd)
        x1 = ops.gemm_nt(att, W[p + "attn.proj.weight"], bias=P[p + "attn.proj.bias"], residual=shortcut, prefetch=W[p + "mlp.layers.0.weight"])
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, P[p + "norm2.weight"], P[p + "norm2.bias"], eps)
        hpre = torch.empty((x1.shape[0], 4 * do), dtype=x.dtype, device=x.device) if save else None
        # bf16: the GEMM saves gelu'(pre-activation) instead of the pre-activation (ACT_GELU_SAVE_GRAD; its exponential is the one the erf
        # already needs) and the backward GEMM multiplies by it (ACT_MUL_H) -- the backward epilogue was bound by re-deriving it (erf + exp
        # per element).  The fp32 parity path keeps the pre-activation.
        # (the modes exist in the bf16 pipelined GEMM kernels: K = do > 64, 8-element-aligned -- every Hiera size, not the 16-wide test trunk)
        save_grad = save and self.dtype == torch.bfloat16 and do > 64 and do % 8 == 0
        act = ops.ACT_GELU_SAVE_GRAD if save_grad else ops.ACT_GELU
        g = ops.gemm_nt(ln2, W[p + "mlp.layers.0.weight"], bias=P[p + "mlp.layers.0.bias"], act=act, preact_out=hpre,
                        prefetch=W[p + "mlp.layers.1.weight"])
        x2 = ops.gemm_nt(g, W[p + "mlp.layers.1.weight"], bias=P[p + "mlp.layers.1.bias"], residual=x1,
                         prefetch=W.get(f"encoder.encoder.blocks.{b['idx'] + 1}.attn.qkv.weight"))
        x2 = x2.view(B, Hq, Wq, do)
        c = None
        if save:
            c = dict(x=x, ln1=ln1, mean1=mean1, rstd1=rstd1, sc_idx=sc_idx, qkv=qkv, qp=qp, q_idx=q_idx, att=att, lse=lse,
                     x1=x1, ln2=ln2, mean2=mean2, rstd2=rstd2, h=hpre, h_is_grad=save_grad, g=g, H=H, W=Wd)
        return x2, c

    def block_bwd(self, b, c, dx2: Tensor, B: int) -> Tensor:
        p = f"encoder.encoder.blocks.{b['idx']}."
        P, W, G = self.P, self.W, self.grad
        dim, do, heads, ws, qs = b["dim"], b["dim_out"], b["heads"], b["window"], b["q_stride"]
        hd = do // heads
        H, Wd = c["H"], c["W"]
        dx2 = dx2.reshape(-1, do)
        self._wg_jobs = [] if self.group_wgrads else None
        # MLP
        dh = self.lin_bwd(p + "mlp.layers.1", dx2, c["g"], gelu_h=c["h"], h_is_grad=c["h_is_grad"], prefetch=p + "mlp.layers.0")
        dln2 = self.lin_bwd(p + "mlp.layers.0", dh, c["ln2"], prefetch=p + "attn.proj")
        dx1 = self.ln_bwd(dln2, c["x1"], P[p + "norm2.weight"], c["mean2"], c["rstd2"], G(p + "norm2.weight"),
                          G(p + "norm2.bias"), dres=dx2)
        # attention branch
        datt = self.lin_bwd(p + "attn.proj", dx1, c["att"], prefetch=p + "attn.qkv")
        dqkv, dqp = ops.attn_bwd(c["qkv"], W[p + "attn.qkv.bias"], c["att"], datt, c["lse"], G(p + "attn.qkv.bias"), B, H, Wd,
                                 heads, hd, ws, q_pooled=c["qp"])
        if qs:
            ops.maxpool2_bwd(dqp, c["q_idx"], dqkv, B, H, Wd, do, 3 * do, 0)
        dln1 = self.lin_bwd(p + "attn.qkv", dqkv.view(-1, 3 * do), c["ln1"],
                            prefetch=f"encoder.encoder.blocks.{b['idx'] - 1}.mlp.layers.1" if b["idx"] > 0 else None)
        if dim != do:
            dsc = torch.empty((B, H, Wd, do), dtype=dx1.dtype, device=dx1.device)
            ops.maxpool2_bwd(dx1, c["sc_idx"], dsc, B, H, Wd, do, do, 0)
            dln1 = self.lin_bwd(p + "proj", dsc.view(-1, do), c["ln1"], residual=dln1)
            dres = None
        else:
            dres = dx1
        if self._wg_jobs is not None:
            jobs, self._wg_jobs = self._wg_jobs, None
            # the groups' small slab reduces are deferred as well (flush_ln_params folds them, 6 per launch) unless a per-unit
            # gradient callback needs finished ranges mid-backward
            self.queue_block_wgrads(jobs)
        dx = self.ln_bwd(dln1, c["x"], P[p + "norm1.weight"], c["mean1"], c["rstd1"], G(p + "norm1.weight"),
                         G(p + "norm1.bias"), dres=dres)
        return dx.view(B, H, Wd, dim)

    def trunk_bwd(self, ctx, dfeats: List[Optional[Tensor]]):
        """dfeats: gradients w.r.t. the 4 stage maps (NHWC, compute dtype) or None."""
        self.trunk_bwd_begin(ctx, dfeats)
        self.trunk_bwd_blocks(0, len(self.blocks))
        self.trunk_bwd_end()

    # The same backward in three callable pieces, so a caller can capture it as several hipGraph segments and start the
    # gradient all-reduce of finished ranges between them (engine/trainer.py, multi-GPU graph mode).
    def trunk_bwd_begin(self, ctx, dfeats: List[Optional[Tensor]]):
        self._bw = dict(ctx=ctx, dfeats=dfeats, dx=None, stage=len(dfeats) - 1, unit=0)
        # (a backward that raised half-way must not leave its deferred weight gradients / LayerNorm jobs / slab reduces to the next one)
        self._wg_pending, self._tn_defer, self._ln_jobs = {}, [], []
        self._sq_parts, self._sq_cover = [], []
        if self.unit_cb is not None:
            self.unit_cb(0)   # head gradients are final once trunk backward starts

    def trunk_bwd_blocks(self, lo: int, hi: int):
        """Back-propagates through blocks hi-1, hi-2, ..., lo (must be called with descending, contiguous ranges)."""
        st = self._bw
        ctx, B = st["ctx"], st["ctx"]["B"]
        for idx in range(hi - 1, lo - 1, -1):
            b, c = self.blocks[idx], ctx["blocks"][idx]
            if b["stage_end"]:
                d = st["dfeats"][st["stage"]]
                st["stage"] -= 1
                if d is not None:
                    st["dx"] = d if st["dx"] is None else ops.add(st["dx"], d.contiguous())
            if st["dx"] is None:
                raise RuntimeError("trunk_bwd: no gradient reaches the last stage")
            st["dx"] = self.block_bwd(b, c, st["dx"], B)
            st["unit"] += 1
            if self.unit_cb is not None:
                self.unit_cb(st["unit"])
        self.join_wgrad()          # (before the deferred slab reduces: they read what forked launches wrote)
        self.flush_ln_params()
        # every gradient of the blocks walked so far is complete here: a caller (the multi-GPU segmented step) all-reduces their range next
        assert not self._wg_pending and not self._ln_jobs, "trunk_bwd_blocks must end with every deferred weight gradient issued"

    def trunk_bwd_end(self):
        e = "encoder.encoder."
        st = self._bw
        ctx, dx = st["ctx"], st["dx"]
        D = dx.shape[-1]
        d2 = dx.reshape(-1, D)
        pw = ops.zeros_f32((D, PATCH_KPAD), dx.device)
        ops.gemm_tn(d2, ctx["cols"], pw, dbias=self.grad(e + "patch_embed.proj.bias"))
        n_b, n_w, Kp = self._basis_dims
        gpos = ops.zeros_f32((D, Kp), dx.device)
        ops.gemm_tn(d2, ctx["basis"], gpos)
        ops.add_cols_batch([(self.grad(e + "patch_embed.proj.weight").view(D, 147), pw), (self.grad(e + "pos_embed").view(D, n_b), gpos),
                            (self.grad(e + "pos_embed_window").view(D, n_w), gpos[:, n_b:])])
        self._bw = None
        self.join_wgrad()

    # ================================================================================================ head
    def conv3_fwd(self, name: str, x: Tensor, B, H, W, Ci, bias: bool):
        return ops.gemm_nt(x, self.W[name + ".weight"], bias=self.P[name + ".bias"] if bias else None, conv=(B, H, W, Ci))

    def conv3_fwd_stats(self, name: str, x: Tensor, B, H, W, Ci, bias: bool, training: bool):
        """conv3_fwd for a convolution that feeds a train-mode BatchNorm: (output, partial statistics or None).  With an instance of the
        halo-tile kernel for the shape, the statistics come out of the convolution's own epilogue."""
        w = self.W[name + ".weight"]
        rows = ops.conv3x3_stats_rows(x, B, H, W, Ci, w.shape[0]) if training else 0
        if rows <= 0:
            return self.conv3_fwd(name, x, B, H, W, Ci, bias), None
        return ops.conv3x3_fwd_stats(x, w, self.P[name + ".bias"] if bias else None, B, H, W, Ci, rows)

    def conv3_bwd(self, name: str, dy: Tensor, x: Tensor, B, H, W, Ci, Co, bias: bool, need_dx: bool = True):
        gw, gb = self.grad(name + ".weight"), (self.grad(name + ".bias") if bias else None)

        def launch():
            if ops.conv3x3_wgrad_direct(dy, x, gw, gb, (B, H, W, Ci)):     # halo-tile kernel, straight into the [Co,Ci,3,3] gradient
                return
            gp = ops.zeros_f32((Co, 9 * Ci), dy.device)
            ops.gemm_tn(dy, x, gp, conv=(B, H, W, Ci), dbias=gb)
            ops.unpack_conv3x3_grad(gp, gw)
        self._wgrad(launch, dy, x)
        if not need_dx:
            return None
        return ops.gemm_nt(dy, self.W[name + ".weight:dgrad"], conv=(B, H, W, Co))

    def head_fwd(self, feats: List[Tensor], training: bool, save: bool):
        """feats: [s2,s3,s4] NHWC.  Returns dict of NHWC outputs + ctx."""
        P, W, T = self.P, self.W, self.dtype
        s2, s3, s4 = feats
        B, h, w, C2 = s2.shape
        C3, C4 = s3.shape[-1], s4.shape[-1]
        HW = h * w
        M = B * HW
        Ct = C2 + C3 + C4
        dev = s2.device
        c: dict = {"B": B, "h": h, "w": w, "chans": (C2, C3, C4), "s_shapes": (s3.shape, s4.shape)}
        # --- CFI fusion: 1x1 conv over cat[s2, up(s3), up(s4)] evaluated per source resolution (a 1x1 conv commutes with bilinear
        # interpolation): three GEMMs + one combine pass, no 2016-channel concat buffer -> BN -> ReLU -> SE
        h3, w3, h4, w4 = s3.shape[1], s3.shape[2], s4.shape[1], s4.shape[2]
        y2 = ops.gemm_nt(s2.reshape(M, C2), W[FUSION_W + ":s2"])
        y3 = ops.gemm_nt(s3.reshape(-1, C3), W[FUSION_W + ":s3"])
        y4 = ops.gemm_nt(s4.reshape(-1, C4), W[FUSION_W + ":s4"])
        f0 = ops.cfi_combine(y2, y3, y4, B, h, w, h3, w3, h4, w4, 512)
        f1, c["bn_f"] = self.bn_fwd("fusion.bn.", f0, 512, True, training, save)
        gap = ops.gap_sum(f1, B, HW, 512)            # column SUMS: the SE kernels take the 1 / HW of the mean as in_scale
        hidden, scale = ops.se_fc(gap, P["fusion.se_block.fc.0.weight"], P["fusion.se_block.fc.2.weight"], 1.0 / HW)
        fused = ops.chan_scale(f1, scale, B, HW, 512)
        # --- e-ASPP
        r0 = ops.gemm_nt(fused, W["context.reduce.0.weight"])
        r1, c["bn_r"] = self.bn_fwd("context.reduce.1.", r0, 128, True, training, save)
        # e-ASPP middle, branch-batched (csrc/easpp.hip): the four dilated depth-wise branches share ONE tensor dcat [M, 512] in the
        # reference's branch-major concat order; their BatchNorm statistics come from one reduction, their BN-apply + ReLU is folded into
        # the grouped 1x1 fusion conv, and the global branch is one single-workgroup kernel
        wd4 = [P[f"context.branches.{i}.0.weight"].view(128, 9) for i in range(4)]
        dcat = ops.dwconv4(r1, wd4, EASPP_RATES, B, h, w, 128)
        bnn = [f"context.branches.{i}.1." for i in range(4)]
        if training:
            ss_b, mi_b = ops.bn_stats_finalize4(dcat, 512, [P[n + "weight"] for n in bnn], [P[n + "bias"] for n in bnn],
                                                [P[n + "running_mean"] for n in bnn], [P[n + "running_var"] for n in bnn],
                                                [P[n + "num_batches_tracked"] for n in bnn])
        else:
            fin = [ops.bn_finalize(None, P[n + "weight"], P[n + "bias"], P[n + "running_mean"], P[n + "running_var"], M, False) for n in bnn]
            ss_b = torch.cat([f[0][:128] for f in fin] + [f[0][128:] for f in fin])
            mi_b = None
        gs = ops.gap_sum(r1, B, HW, 128)
        gn = "context.global_branch.2."
        gm, gl0, glob, ss_g, mi_g = ops.easpp_global_fwd(gs, P["context.global_branch.1.weight"].view(128, 128), P[gn + "weight"], P[gn + "bias"],
                                                         P[gn + "running_mean"], P[gn + "running_var"], P[gn + "num_batches_tracked"], B, 128, HW,
                                                         training)
        fu0 = ops.easpp_fuse_bn(dcat, ss_b, glob, P["context.fusion.0.weight"].view(128, 5), B, HW, 128)
        fu1, c["bn_u"] = self.bn_fwd("context.fusion.1.", fu0, 128, True, training, save)
        e0 = ops.gemm_nt(fu1, W["context.expand.0.weight"])
        context, c["bn_e"] = self.bn_fwd("context.expand.1.", e0, 256, True, training, save)
        # --- EFE: conv3x3 -> BN statistics -> [BN-apply + ReLU + 1x1 edge head] in one pass
        ec, ec_part = self.conv3_fwd_stats("edge_detector.conv1", context, B, h, w, 256, False, training)
        c["bn_ef"] = st_ef = self.bn_prepare("edge_detector.bn1.", ec, 64, True, training, save, part=ec_part)
        edge_f, edge = ops.bn_apply_head(ec, st_ef.ss, P["edge_detector.edge_conv.weight"].view(64), P["edge_detector.edge_conv.bias"], 64)
        # --- PED.  Per stage: ONE gather builds the conv input cat[up2(relu(bn2(previous raw conv output))), up(edge_features)] (the
        # previous stage's BN-apply is folded into the gather, its activated output is never stored); conv1 -> BN -> ReLU -> conv2 ->
        # BN statistics -> [BN-apply + ReLU + 1x1 prediction head] reading the raw conv2 output once.
        preds, stages = [], []
        x, x_ss, Hc, Wc, Cin = context, None, h, w, 256
        for i, (Co, ec_ch) in enumerate(zip((256, 128, 64), (64, 64, 0))):
            H2, W2 = Hc * 2, Wc * 2
            Cc = Cin + ec_ch
            pc = ops.ped_gather(x, x_ss, B, Hc, Wc, Cin, edge_f if ec_ch else None, h, w, ec_ch)
            pre = f"decoder.decoder_blocks.{i}."
            a0, a0_part = self.conv3_fwd_stats(pre + "conv1", pc, B, H2, W2, Cc, True, training)
            a1, st1 = self.bn_fwd(pre + "bn1.", a0, Co, True, training, save, part=a0_part)
            b0, b0_part = self.conv3_fwd_stats(pre + "conv2", a1, B, H2, W2, Co, True, training)
            st2 = self.bn_prepare(pre + "bn2.", b0, Co, True, training, save, part=b0_part)
            _, pred = ops.bn_apply_head(b0, st2.ss, P[f"decoder.pred_heads.{i}.weight"].view(Co), P[f"decoder.pred_heads.{i}.bias"], Co,
                                        write_y=False)
            preds.append(pred.view(B, 1, H2, W2))
            stages.append(dict(pc=pc if save else None, a1=a1 if save else None, bn1=st1, bn2=st2 if save else None, H=H2, W=W2, Cin=Cin,
                               ec=ec_ch, Co=Co))
            x, x_ss, Hc, Wc, Cin = b0, st2.ss, H2, W2, Co
        out = {"predictions": preds, "edge": edge.view(B, 1, h, w), "context": context.view(B, h, w, 256),
               "fused": fused.view(B, h, w, 512), "edge_features": edge_f.view(B, h, w, 64)}
        if save:
            c.update(s2=s2, s3=s3, s4=s4, f1=f1, gap=gap, hidden=hidden, scale=scale, fused=fused, r1=r1, dcat=dcat, ss_b=ss_b, mi_b=mi_b, gm=gm,
                     gl0=gl0, glob=glob, mi_g=mi_g, fu1=fu1, context=context, edge_f=edge_f, stages=stages)
        return out, (c if save else None)

    def head_bwd(self, c, dpreds: List[Optional[Tensor]], dedge: Optional[Tensor], dextra: Optional[dict] = None):
        """Returns gradients w.r.t. [s2,s3,s4] (NHWC).  dpreds[i]: [B,1,H,W] or None; dedge: [B,1,h,w] or None."""
        P, W, G, T = self.P, self.W, self.grad, self.dtype
        B, h, w = c["B"], c["h"], c["w"]
        HW, M = h * w, c["B"] * c["h"] * c["w"]
        dev = c["f1"].device
        C2, C3, C4 = c["chans"]

        def zeros(*shape):
            return torch.zeros(shape, dtype=T, device=dev)

        d_edge_f = torch.empty((M, 64), dtype=T, device=dev)
        d_context = torch.empty((M, 256), dtype=T, device=dev)
        edge_acc = False
        # --- PED, last stage first.  BN2 backward forms its incoming gradient d_next + dpred (x) w_head on the fly (bn_bwd_head): the
        # head's rank-one dx and the activated stage output are never materialised.
        d_next = None  # gradient w.r.t. this stage's activated output coming from the next stage's gather
        for i in (2, 1, 0):
            st = c["stages"][i]
            H2, W2, Cin, ec_ch, Co = st["H"], st["W"], st["Cin"], st["ec"], st["Co"]
            Mi = B * H2 * W2
            pre = f"decoder.decoder_blocks.{i}."
            bn2 = st["bn2"]
            dp = dpreds[i].to(T).contiguous().view(Mi) if dpreds[i] is not None else torch.zeros(Mi, dtype=T, device=dev)
            d_b0 = ops.bn_bwd_head(d_next, bn2.x, dp, P[f"decoder.pred_heads.{i}.weight"].view(Co), bn2.ss, bn2.mi, P[pre + "bn2.weight"],
                                   G(pre + "bn2.weight"), G(pre + "bn2.bias"), G(f"decoder.pred_heads.{i}.weight").view(Co),
                                   G(f"decoder.pred_heads.{i}.bias"), Co)
            d_a1 = self.conv3_bwd(pre + "conv2", d_b0, st["a1"], B, H2, W2, Co, Co, bias=True)
            d_a0 = self.bn_bwd(st["bn1"], d_a1)
            Cc = Cin + ec_ch
            d_pc = self.conv3_bwd(pre + "conv1", d_a0, st["pc"], B, H2, W2, Cc, Co, bias=True)
            Hc, Wc = H2 // 2, W2 // 2
            if i == 0:
                ops.ped_gather_bwd(d_pc, d_context, B, Hc, Wc, Cin, H2, W2, Cc, 0)
                d_next = None
            else:
                d_next = torch.empty((B * Hc * Wc, Cin), dtype=T, device=dev)
                ops.ped_gather_bwd(d_pc, d_next, B, Hc, Wc, Cin, H2, W2, Cc, 0)
            if ec_ch:
                ops.ped_gather_bwd(d_pc, d_edge_f, B, h, w, 64, H2, W2, Cc, Cin, accumulate=edge_acc)
                edge_acc = True
        have_ctx = True
        # --- EFE: BN backward with the edge head's gradient formed on the fly
        if dextra and dextra.get("edge_features") is not None:
            d_edge_f = ops.add(d_edge_f, dextra["edge_features"].to(T).contiguous().view(M, 64))
        bn_ef = c["bn_ef"]
        de = dedge.to(T).contiguous().view(M) if dedge is not None else torch.zeros(M, dtype=T, device=dev)
        d_ec = ops.bn_bwd_head(d_edge_f, bn_ef.x, de, P["edge_detector.edge_conv.weight"].view(64), bn_ef.ss, bn_ef.mi,
                               P["edge_detector.bn1.weight"], G("edge_detector.bn1.weight"), G("edge_detector.bn1.bias"),
                               G("edge_detector.edge_conv.weight").view(64), G("edge_detector.edge_conv.bias"), 64)
        d_ctx2 = self.conv3_bwd("edge_detector.conv1", d_ec, c["context"], B, h, w, 256, 64, bias=False)
        d_context = ops.add(d_context, d_ctx2) if have_ctx else d_ctx2
        if dextra and dextra.get("context") is not None:
            d_context = ops.add(d_context, dextra["context"].to(T).contiguous().view(M, 256))
        # --- e-ASPP
        d_e0 = self.bn_bwd(c["bn_e"], d_context)
        d_fu1 = self.lin_bwd("context.expand.0", d_e0, c["fu1"], bias=False)
        d_fu0 = self.bn_bwd(c["bn_u"], d_fu1)
        # fusion conv + branch BatchNorms + global branch + depth-wise convs, branch-batched (mirrors head_fwd)
        wf, dwf = P["context.fusion.0.weight"].view(640), G("context.fusion.0.weight").view(640)
        S = ops.gap_sum(d_fu0, B, HW, 128)
        gn = "context.global_branch.2."
        gadd = ops.easpp_global_bwd(S, c["glob"], c["gl0"], c["gm"], wf, P["context.global_branch.1.weight"].view(128, 128), P[gn + "weight"],
                                    c["mi_g"], dwf, G("context.global_branch.1.weight").view(128, 128), G(gn + "weight"), G(gn + "bias"), B, 128, HW)
        bnn = [f"context.branches.{i}.1." for i in range(4)]
        d_dcat = ops.easpp_fuse_bn_bwd(d_fu0, c["dcat"], wf, c["ss_b"], c["mi_b"], [P[n + "weight"] for n in bnn], [G(n + "weight") for n in bnn],
                                       [G(n + "bias") for n in bnn], dwf, B, HW, 128)
        wd4 = [P[f"context.branches.{i}.0.weight"].view(128, 9) for i in range(4)]
        ops.dwconv4_wgrad(d_dcat, c["r1"], EASPP_RATES, [G(f"context.branches.{i}.0.weight").view(128, 9) for i in range(4)], B, h, w, 128)
        d_r1 = ops.dwconv4_dgrad(d_dcat, wd4, EASPP_RATES, gadd, B, h, w, 128)
        d_r0 = self.bn_bwd(c["bn_r"], d_r1)
        d_fused = self.lin_bwd("context.reduce.0", d_r0, c["fused"], bias=False)
        if dextra and dextra.get("fused") is not None:
            d_fused = ops.add(d_fused, dextra["fused"].to(T).contiguous().view(M, 512))
        # --- SE + fusion
        dscale = ops.chan_prod_sum(d_fused, c["f1"], B, HW, 512)
        dgap = ops.se_fc_bwd(c["gap"], P["fusion.se_block.fc.0.weight"], P["fusion.se_block.fc.2.weight"], c["hidden"], c["scale"],
                             dscale, G("fusion.se_block.fc.0.weight"), G("fusion.se_block.fc.2.weight"), 1.0 / HW)
        d_f1 = ops.chan_scale_bwd(d_fused, c["scale"], dgap, B, HW, 512)
        d_f0 = self.bn_bwd(c["bn_f"], d_f1)
        # fusion conv backward per source: d_y2 = d_f0, d_y3 / d_y4 = the bilinear adjoints of d_f0 (512 channels at the coarse
        # resolutions), then one dgrad + one wgrad GEMM per source (the wgrads write their column slice of the [512, 2016] gradient)
        s3s, s4s = c["s_shapes"]
        d_y3 = torch.empty((B * s3s[1] * s3s[2], 512), dtype=T, device=dev)
        ops.ped_gather_bwd(d_f0, d_y3, B, s3s[1], s3s[2], 512, h, w, 512, 0)
        d_y4 = torch.empty((B * s4s[1] * s4s[2], 512), dtype=T, device=dev)
        ops.ped_gather_bwd(d_f0, d_y4, B, s4s[1], s4s[2], 512, h, w, 512, 0)
        gw = G(FUSION_W).view(512, C2 + C3 + C4)
        outs = []
        for tag, dyk, xk in ((":s2", d_f0, c["s2"]), (":s3", d_y3, c["s3"]), (":s4", d_y4, c["s4"])):
            a, b = self.fusion_split()[tag]
            xk2 = xk.reshape(-1, b - a)
            self._wgrad(lambda dyk=dyk, xk2=xk2, a=a, b=b: ops.gemm_tn(dyk, xk2, gw[:, a:b]), dyk, xk2)
            outs.append(ops.gemm_nt(dyk, W[FUSION_W + tag + "T"]))
        d_s2, d_s3, d_s4 = outs[0].view(B, h, w, C2), outs[1].view(s3s), outs[2].view(s4s)
        return [d_s2, d_s3, d_s4]
