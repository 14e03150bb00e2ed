"""Data-parallel gradient synchronisation: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI.  The reference is single-GPU (no distributed code at all); SURVEY.md §8(e) defines this row.

The image batch shards naturally: every rank runs the same step on its own images (BatchNorm statistics stay
per-rank, as in stock DDP).  The only collective is the gradient sum: the flat gradient arena (engine/arena.py)
is ordered by backward-completion, so as soon as the head / a trunk block has finished its backward, the
contiguous range that just became final is all-reduced on a side stream while the next block's backward runs on
the compute stream.  Buckets are >= `bucket_mb` so each collective is large enough to spread over the 7 xGMI
links; the optimizer then applies 1/world_size through its grad_scale (no extra pass over the gradients).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_process_group_from_env(device_type: str = "cuda") -> Tuple[int, int, int]:
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SPG_DIST_FORCE_INIT=1: a process group even for one rank (tests/test_distributed_gpu.py runs the N > 1 step's choreography -- RCCL
    # collectives between captured graph segments -- on the one GPU a test box has; see GradSync(force=True))
    if (world > 1 or os.environ.get("SPG_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's resident all-reduce blocks and the persistent GEMM kernels (which fill a CU completely) would fight for CUs while a
        # collective overlaps the backward pass: RCCL is capped to 16 channels here and the trainer sizes the GEMM grids that run
        # beside a collective to COMM_CU_BUDGET CUs (engine/trainer.py; the cu_budget argument of the GEMM entry points).  The step's
        # gradients (431 MB as bf16, 862 MB as fp32: GradSync's payload type) have > 15 ms of backward to hide in, so 16 channels are plenty.
        # Both numbers are defaults chosen without an 8-GPU measurement (none was available): set the variable to override.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        # "nccl" IS RCCL on ROCm.  SPG_DIST_BACKEND=gloo exists only to rehearse the N>1 code path with several ranks on ONE GPU
        backend = os.environ.get("SPG_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def graph_capture_mode() -> str:
    """capture_error_mode for torch.cuda.graph.  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events of
    outstanding collectives (hipEventQuery); under the default "global" mode any such call by any thread while a stream captures is an
    error -- the watchdog dies with "operation not permitted when stream is capturing" and takes the process with it (found by the
    one-rank RCCL rehearsal, tests/test_distributed_gpu.py; gloo has no such thread).  "thread_local" restricts the check to the
    capturing thread."""
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


# CUs left to the GEMM grids while a gradient all-reduce is in flight (the other 16 go to RCCL's channels)
COMM_CU_BUDGET = int(os.environ.get("SPG_COMM_CUS", "240"))


def make_buckets(unit_ends: List[int], bucket_elems: int) -> List[Tuple[int, int]]:
    """Greedy grouping of consecutive units into buckets of at least `bucket_elems` elements (last may be smaller)."""
    buckets, start = [], 0
    for e in unit_ends:
        if e - start >= bucket_elems:
            buckets.append((start, e))
            start = e
    if start < unit_ends[-1]:
        buckets.append((start, unit_ends[-1]))
    return buckets


class GradSync:
    """Gradient all-reduce over the flat arena.  compress_bf16=True sends bf16 (431 MB instead of 862 MB per step over xGMI;
    the sum of <= 8 ranks is formed in bf16 by RCCL, the fp32 gradient is rounded once going in and widened coming out); False keeps fp32
    on the wire.  The Trainer chooses: `training.grad_allreduce_dtype` ('bf16' | 'fp32'), default = the compute dtype, so the fp32 parity
    mode never rounds its gradients silently."""

    def __init__(self, grad_flat: torch.Tensor, unit_ends: List[int], bucket_mb: float = 48.0, group=None,
                 compress_bf16: bool = True, force: bool = False):
        self.g = grad_flat
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # collectives are issued when there is more than one rank -- or when `force` asks for them on a one-rank group (a rehearsal: the
        # sum over one rank is the identity, but every RCCL call, stream hand-over and staging cast of the N > 1 step executes)
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.group = group
        self.buckets = make_buckets(unit_ends, int(bucket_mb * 1024 * 1024 / 4))
        self.compress = compress_bf16
        self.cuda = grad_flat.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self._next = 0
        self._works = []
        self._tmp = {}
        self._sq = {}        # (s, e) -> partial sums of squares of the all-reduced range, written by the cast-back pass (bf16 wire)

    def fold_for(self, ranges):
        """(partial arrays, gradient views) for Arena.step(fold=...): the sums of squares of the all-reduced gradient ranges that the
        bf16 wire's cast-back pass wrote while it touched every element anyway (reduce_range), so the clip's norm pass reads only what
        these ranges do not cover.  None when a range went over the wire as fp32 (no cast, no partials).  Buffers persist per range:
        a captured optimizer graph may hold their addresses."""
        if not (self.active and self.compress and self.cuda) or os.environ.get("SPG_SYNC_FOLD", "1") == "0":     # (the switch: A/B runs)
            return None
        parts, views = [], []
        for s, e in ranges:
            if (e - s) % 8 != 0 or e <= s:
                return None
            parts.append(self._sq_buf(s, e))
            views.append(self.g[s:e])
        return (parts, views) if 0 < len(parts) <= 32 else None

    def _sq_buf(self, s, e):
        buf = self._sq.get((s, e))
        if buf is None:
            n = max(1, min(1024, (e - s) // (8 * 256 * 4)))
            buf = self._sq[(s, e)] = torch.zeros(n, dtype=torch.float32, device=self.g.device)
        return buf

    def reset(self):
        self._next = 0
        self._works = []

    def ready(self, upto: int):
        """Gradients in [0, upto) are final: launch every not-yet-launched bucket that ends at or before `upto`."""
        if not self.active:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= upto:
            s, e = self.buckets[self._next]
            self._next += 1
            self._launch(s, e)

    def drop_staging(self):
        """Frees the bf16 staging buffers (a warm-up that reduced the whole arena at once leaves a 431 MB one the segmented replay never uses)."""
        self._tmp = {}

    def reduce_range(self, s: int, e: int):
        """all-reduce (sum) of g[s:e] on the CURRENT stream; bf16 payload through a persistent staging buffer (HIP cast kernels)"""
        view = self.g[s:e]
        if self.compress and self.cuda and (e - s) % 8 == 0:
            from .. import _lib
            tmp = self._tmp.get((s, e))
            if tmp is None:
                tmp = self._tmp[(s, e)] = torch.empty(e - s, dtype=torch.bfloat16, device=view.device)
            st = torch.cuda.current_stream().cuda_stream
            _lib.call("spg_cast_bf16", view.data_ptr(), tmp.data_ptr(), e - s, 0, st)
            dist.all_reduce(tmp, group=self.group)
            sq = self._sq_buf(s, e)
            _lib.call("spg_cast_bf16_sq", view.data_ptr(), tmp.data_ptr(), e - s, sq.data_ptr(), sq.numel(), st)
        else:
            dist.all_reduce(view, group=self.group)

    def _launch(self, s: int, e: int):
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.reduce_range(s, e)
        else:
            self._works.append(dist.all_reduce(self.g[s:e], group=self.group, async_op=True))

    def finish(self):
        """All buckets launched and complete w.r.t. the current stream; returns the grad scale 1/world."""
        if self.active:
            self.ready(self.buckets[-1][1])
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
            else:
                for w in self._works:
                    w.wait()
        self.reset()
        return 1.0 / self.world
