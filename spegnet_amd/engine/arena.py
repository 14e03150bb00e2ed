"""Flat fp32 parameter / gradient arena and the fused optimizer step over it.

Parameters are laid out in the order their gradients become final during backward (head first, then trunk
blocks last-to-first, then patch/pos embedding), each starting on a 256-element boundary.  That gives
  * one memset to clear gradients, one kernel for the global grad norm, one kernel for AdamW;
  * contiguous, in-order-ready ranges for the bucketed RCCL all-reduce (engine/distributed.py).
Group rules follow Trainer._get_param_groups (reference engine/trainer.py:274-306), including its quirk that BN
layers inside nn.Sequential (no 'bn'/'norm' in the name) do receive weight decay.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import torch

from .. import _lib

ALIGN = 256


def backward_order(names: List[str]) -> List[str]:
    def key(n: str):
        if not n.startswith("encoder."):
            return (0, 0)
        m = re.search(r"blocks\.(\d+)\.", n)
        if m:
            return (1, -int(m.group(1)))
        return (2, 0)
    return sorted(names, key=key)  # stable: keeps definition order inside a unit


def group_of(name: str) -> int:
    norm = ('norm' in name) or ('bn' in name)
    if 'encoder' in name:
        return 1 if norm else 0
    return 3 if norm else 2


class Arena:
    def __init__(self, model: torch.nn.Module):
        params: Dict[str, torch.nn.Parameter] = dict(model.named_parameters())
        self.names = backward_order(list(params.keys()))
        dev = next(iter(params.values())).device
        offs, off = {}, 0
        for n in self.names:
            offs[n] = off
            off += (params[n].numel() + ALIGN - 1) // ALIGN * ALIGN
        self.size = off
        self.offsets = offs
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        groups = torch.zeros(off // ALIGN, dtype=torch.uint8)
        for n in self.names:
            prm = params[n]
            k = prm.numel()
            o = offs[n]
            self.p[o:o + k].copy_(prm.data.reshape(-1))
            prm.data = self.p[o:o + k].view(prm.shape)
            prm.grad = self.g[o:o + k].view(prm.shape)
            groups[o // ALIGN:(o + k + ALIGN - 1) // ALIGN] = group_of(n)
        self.group_of_chunk = groups.to(dev)
        # unit boundaries (end offsets) in readiness order: head, block L-1, ..., block 0, embeddings
        self.unit_ends: List[int] = []
        prev_key = None
        for n in self.names:
            m = re.search(r"blocks\.(\d+)\.", n) if n.startswith("encoder.") else None
            key = ("head",) if not n.startswith("encoder.") else (("blk", m.group(1)) if m else ("emb",))
            if prev_key is not None and key != prev_key:
                self.unit_ends.append(offs[n])
            prev_key = key
        self.unit_ends.append(off)
        self.m = self.v = None
        self._clean = True   # arena.g is all zeros
        self._unzeroed = frozenset()   # ... except the matrices at these offsets (left uncleared by the last optimizer step, see zero_grad)
        self.step_f = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr = torch.zeros(4, dtype=torch.float32, device=dev)
        self.wd = torch.zeros(4, dtype=torch.float32, device=dev)

    def reattach(self, model: torch.nn.Module):
        """Restore .grad views (e.g. after optimizer.zero_grad(set_to_none=True))."""
        for n, prm in model.named_parameters():
            o, k = self.offsets[n], prm.numel()
            prm.grad = self.g[o:o + k].view(prm.shape)

    def set_hyper(self, base_lr: float, weight_decay: float, encoder_lr_ratio: float):
        self.lr.copy_(torch.tensor([base_lr * encoder_lr_ratio, base_lr * encoder_lr_ratio, base_lr, base_lr]))
        self.wd.copy_(torch.tensor([0.0, 0.0, weight_decay, 0.0]))

    def scale_lr(self, factor: float, min_lr: float):
        self.lr.copy_(torch.clamp(self.lr * factor, min=min_lr))

    def zero_grad(self, expect_overwrite=None):
        """Clears the gradient arena unless the previous optimizer step already did (adamw zero_grad=1).  The previous step may have left
        the gradients of some matrices uncleared (`_unzeroed`: arena offsets) on the caller's promise that the next backward stores them
        whole; expect_overwrite repeats that promise for this backward (TrainStep: same shapes, same switches) -- without it they are
        cleared here."""
        if not self._clean or (self._unzeroed and expect_overwrite != self._unzeroed):
            self.g.zero_()
            self._unzeroed = frozenset()
        self._clean = False

    def _fold_plan(self, cover):
        """Chunk table (device) of the gradient arena MINUS the tensors in `cover` (views of self.g written by spg_gemm_tn_blocks launches
        that also produced their sums of squares): records struct { long off; int n4; int pad; }, <= 16384 floats each.  Cached per set."""
        import struct
        base = self.g.data_ptr()
        rng = sorted(((t.data_ptr() - base) // 4, t.numel()) for t in cover)
        key = tuple(rng)
        if getattr(self, "_fold_key", None) == key:
            return self._fold_blob, self._fold_n
        recs, pos = [], 0
        for off, n in rng + [(self.size, 0)]:
            if off < pos or off % 4 or n % 4:
                raise RuntimeError("fold plan: covered gradient ranges overlap or are not 16-byte aligned")
            a = pos
            while a < off:                       # the uncovered stretch [pos, off), in chunks of <= 16384 floats
                m = min(16384, off - a)
                recs.append(struct.pack("<qii", a, m // 4, 0))
                a += m
            pos = off + n
        if not recs:
            recs.append(struct.pack("<qii", 0, 0, 0))
        self._fold_blob = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(self.p.device)
        self._fold_key, self._fold_n = key, len(recs)
        return self._fold_blob, self._fold_n

    def step(self, clip: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0, packer=None, fold=None, keep=None):
        """Global-norm clip + AdamW.  packer (the model's Engine): the update kernel also writes the compute-dtype weight copies of the
        next forward (spg_adamw_pack) instead of leaving them to a separate re-pack pass.  fold = (sums of squares, covered gradient
        tensors) from Engine.take_sq(): the norm pass then reads only the uncovered gradients (spg_sumsq_fold).  keep = gradient tensors
        (views of self.g) this backward STORED whole: the matrices among them are not cleared -- the caller promises that its next
        backward stores them whole again (zero_grad(expect_overwrite)).  keep works without fold (multi-GPU: the norm is that of the
        all-reduced gradients, a full pass)."""
        if self.m is None:
            self.m = torch.zeros_like(self.p)
            self.v = torch.zeros_like(self.p)
        s = torch.cuda.current_stream().cuda_stream
        from .. import ops
        if getattr(self, "_red_ws", None) is None:
            self._red_ws = torch.empty(2048, dtype=torch.float32, device=self.p.device)
        es = 2 if (packer is not None and packer.dtype == torch.bfloat16) else 4
        use_fold = fold is not None and len(fold[0]) > 0 and len(fold[0]) <= 32       # (spg_sumsq_fold takes up to 32 partial arrays)
        base = self.g.data_ptr()
        stored = keep if keep is not None else (fold[1] if fold is not None else [])
        cov_offsets = frozenset((t.data_ptr() - base) // 4 for t in stored)
        if self._unzeroed and not self._unzeroed <= cov_offsets:
            raise RuntimeError("gradients left uncleared by the previous optimizer step were not overwritten by this backward (stale values): "
                               "call zero_grad() without a promise before a backward that makes other launches")
        # algorithmic bytes: sumsq reads g; adamw reads p, g, m, v and writes p, m, v, g (cleared) (+ the two compute-dtype copies); with a
        # fold the covered gradients are neither read by the norm pass nor cleared (8 bytes per covered element less)
        covered = sum(t.numel() for t in fold[1] if t.dim() == 2) if use_fold else 0
        kept = sum(t.numel() for t in keep if t.dim() == 2) if (keep is not None and packer is not None) else 0
        with ops._prof("sumsq + adamw_pack (clip + AdamW + weight re-pack)" if packer is not None else "sumsq + adamw", "hbm",
                       self.size * (4 + 32 + (2 * es if packer is not None else 0)) - 4 * covered - 4 * kept):
            # the stored MATRICES stay uncleared: the same launches store them whole next step (biases are added to)
            keep_offs = frozenset((t.data_ptr() - base) // 4 for t in keep if t.dim() == 2) if (keep is not None and packer is not None) else frozenset()
            if use_fold:
                import ctypes
                blob, nchunks = self._fold_plan(fold[1])
                if self._red_ws.numel() < nchunks:
                    self._red_ws = torch.empty(max(nchunks, 2048), dtype=torch.float32, device=self.p.device)
                ne = len(fold[0])
                EP, EI = ctypes.c_void_p * ne, ctypes.c_int * ne
                _lib.call("spg_sumsq_fold", self.g.data_ptr(), blob.data_ptr(), nchunks, ne, EP(*[t.data_ptr() for t in fold[0]]),
                          EI(*[t.numel() for t in fold[0]]), self.gnorm_sq.data_ptr(), self._red_ws.data_ptr(), self._red_ws.numel(),
                          ops.red_counters(self.p.device, 1), s)
            else:
                _lib.call("spg_sumsq", self.g.data_ptr(), self.gnorm_sq.data_ptr(), self.size, self._red_ws.data_ptr(), 2048,
                          ops.red_counters(self.p.device, 1), s)
            if packer is not None:
                blob, njobs, items = packer.opt_jobs(self, keep_offs)
                dt = _lib.SPG_BF16 if packer.dtype == torch.bfloat16 else _lib.SPG_F32
                _lib.call("spg_adamw_pack", dt, self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                          self.group_of_chunk.data_ptr(), self.lr.data_ptr(), self.wd.data_ptr(), self.gnorm_sq.data_ptr(),
                          self.step_f.data_ptr(), float(clip), betas[0], betas[1], eps, float(grad_scale), 1, blob.data_ptr(), njobs, items, s)
            else:
                _lib.call("spg_adamw", self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                          self.group_of_chunk.data_ptr(), self.lr.data_ptr(), self.wd.data_ptr(), self.gnorm_sq.data_ptr(),
                          self.step_f.data_ptr(), float(clip), betas[0], betas[1], eps, float(grad_scale), 1, self.size, s)
        self._clean = True
        self._unzeroed = keep_offs

    def state_dict(self):
        """Optimizer state of the flat arena: Adam moments (arena layout), step counter, per-group lr / weight decay, and the name ->
        (offset, numel) table that lets load_state_dict verify the layout."""
        if self.m is None:
            self.m, self.v = torch.zeros_like(self.p), torch.zeros_like(self.p)
        return {"m": self.m.clone(), "v": self.v.clone(), "step": self.step_f.clone(), "lr": self.lr.clone(), "wd": self.wd.clone(),
                "layout": {n: self.offsets[n] for n in self.names}, "size": self.size}

    def load_state_dict(self, sd) -> None:
        """Restores what state_dict() saved (checkpoint resume: the reference only saves, engine/trainer.py:588-606).  The parameters
        themselves come from the model's state_dict; this restores m, v, the step counter and the learning rates."""
        if int(sd["size"]) != self.size or any(self.offsets.get(n) != o for n, o in sd["layout"].items()):
            raise ValueError("optimizer state was saved for a different parameter layout")
        dev = self.p.device
        self.m = sd["m"].to(dev, torch.float32).clone()
        self.v = sd["v"].to(dev, torch.float32).clone()
        self.step_f.copy_(sd["step"].to(dev))
        self.lr.copy_(sd["lr"].to(dev))
        self.wd.copy_(sd["wd"].to(dev))
