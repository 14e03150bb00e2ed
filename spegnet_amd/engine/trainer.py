"""Training entry points for the MI355X path, mirroring the reference's engine/trainer.py surface:
`Trainer(config, dir_manager, device)`, `.train(dataset_dirs)`, `.train_epoch(loader, epoch)`,
`.validate(loader)`, `._process_batch(batch, is_train)` (reference engine/trainer.py:201-606).

What differs by design (MI355X-first):
  * mixed precision is bf16 compute with fp32 master weights and NO GradScaler (`use_amp: true` -> bf16,
    false -> fp32 parity mode); the reference's fp16 autocast + GradScaler is a CUDA policy, not the algorithm;
  * optimizer = fused global-norm clip + AdamW over a flat arena with the reference's 4 parameter groups;
  * the fixed-shape step can be captured into one hipGraph (`TrainStep(capture=True)`);
  * data parallelism over RCCL is built in (engine/distributed.py) -- the reference is single-GPU.
"""
from __future__ import annotations

import json
import logging
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .. import ops
from ..models.spegnet import SPEGNet
from ..utils.loss_functions import CODLoss
from .arena import Arena
from .distributed import COMM_CU_BUDGET, GradSync, graph_capture_mode

logger = logging.getLogger(__name__)


class TrainStep:
    """One optimisation step on fixed-shape device tensors: forward, CODLoss, backward, (all-reduce), clip+AdamW,
    weight re-pack.

    capture=False : eager launches; with a GradSync the bucketed all-reduce overlaps the trunk backward.
    capture=True  : the step is recorded once into hipGraphs and replayed (removes ~1.5k host launches per step).
                    world==1: one graph for the whole step.
                    world>1 : the backward is written out by hand (no autograd inside) and captured as SEGMENTS --
                    [forward + loss + head backward + last trunk blocks], [next blocks], ..., [first blocks + embeddings] --
                    after each segment's replay the gradient range that just became final is all-reduced over RCCL on
                    a side stream while the next segment runs; a last graph does clip + AdamW + re-pack.  Collectives
                    stay outside graph capture, yet overlap with backward."""

    def __init__(self, model: SPEGNet, criterion: CODLoss, arena: Arena, grad_clip: float = 1.0, sync: Optional[GradSync] = None,
                 capture: bool = False, force_segmented: bool = False, wgrad_async: bool = False):
        self.model, self.criterion, self.arena, self.clip, self.sync = model, criterion, arena, grad_clip, sync
        self.capture = capture
        self.force_segmented = force_segmented   # tests: exercise the multi-GPU segmented capture on one rank
        self.graph = self.graph_b = None
        self.segments = None
        self.comm_stream = None
        # graph segments of the backward (the all-reduce of each finished range overlaps the next segment); SPG_SEGMENTS: A/B runs
        # (six: eight blocks per segment -- under the 240-CU budget a segment's stage-3 weight gradients then pack into whole-block launches
        # without a remainder for the tile kernel, and every seam between two graph launches costs ~0.1 ms: 1 / 2 / 4 / 6 / 8 / 16 segments
        # = 21.3 / 21.8 / 22.0 / 22.4 / 22.65 / 22.8 ms against 20.8-21.2 for the single graph, tools/seg_probe.py; the last range -- the
        # only exposed all-reduce -- stays small: blocks 0-7 are stages 1 and 2, 7 M parameters)
        self.n_segments = int(os.environ.get("SPG_SEGMENTS", "6"))
        self.static = None
        self.losses = None
        self.world = sync.world if sync is not None else 1
        self.comm = sync is not None and sync.active     # collectives are issued (more than one rank, or a forced one-rank rehearsal)
        self.fold_sumsq = os.environ.get("SPG_FOLD_SUMSQ", "1") != "0"     # (A/B runs: tools/ab_env.sh)
        self._graph_unzeroed = frozenset()
        # forked weight-gradient stream (models/engine.py; `TrainStep(..., wgrad_async=True)` for A/B runs): measured SLOWER on one MI355X
        # (226.0 vs 232.2 img/s on the B=8@384 graph step) -- both branches are chip-sized persistent kernels, so they contend
        model.engine.wgrad_async = ((sync is None) or capture) and bool(wgrad_async)
        if sync is not None and not capture:
            ends = arena.unit_ends
            model.engine.unit_cb = lambda k: sync.ready(ends[k])
            model.cu_budget = COMM_CU_BUDGET      # eager: bucket all-reduces overlap the whole backward (applied per call, see SPEGNet)

    # ---- pieces -------------------------------------------------------------------------------------------
    def _fwd_bwd(self, images, masks, edges):
        ops.begin_zero_pool(images.device)   # one fill for the step's zero-initialised scratch (ops.zeros_f32)
        # (the matrices the previous step left uncleared are stored whole again by this backward IF it makes the same launches: same input
        # shapes, same switches -- otherwise they are cleared now; Arena.step refuses a backward that broke the promise)
        self._begin_grads(images)
        out = self.model(images)
        losses = self.criterion.forward_batched(out['predictions'], out['edge'], masks, edges)
        if getattr(self, "_one", None) is None:       # the root gradient, kept: backward() without it fills a fresh ones_like every step
            self._one = torch.ones((), dtype=torch.float32, device=images.device)
        losses['loss'].backward(self._one)
        return {k: v.detach() for k, v in losses.items()}

    def _begin_grads(self, images):
        """Start of a backward's gradient bookkeeping.  The whole-block weight-gradient launches STORE their blocks (the optimizer then
        leaves those matrices uncleared: they are stored whole again by the next backward IF it makes the same launches -- same input
        shapes, same switches; otherwise they are cleared now, and Arena.step refuses a backward that broke the promise).  Single GPU: the
        norm is taken right after the backward, so the launches also hand their sums of squares to the clip (models/engine.py:
        fold_sumsq).  With a GradSync the norm is that of the all-reduced gradients: stored and kept, but no fold."""
        sig = (tuple(images.shape), self.fold_sumsq, self.sync is None)
        self.arena.zero_grad(expect_overwrite=self.arena._unzeroed if sig == getattr(self, "_fold_sig", None) else None)
        self._fold_sig = sig
        eng = self.model.engine
        eng.fold_sumsq = self.sync is None and self.fold_sumsq
        eng.store_wgrads = self.fold_sumsq

    def _opt(self, scale: float, reduced=None):
        """reduced: the gradient ranges [(s, e)] this step all-reduced (their sums of squares come from the wire's cast-back pass)"""
        # AdamW writes the compute-dtype weight copies of the next forward itself (spg_adamw_pack): no separate re-pack pass
        eng = self.model._engine
        parts, cover = eng.take_sq()
        fold = (parts, cover) if eng.fold_sumsq else None
        if fold is None and self.comm and reduced:
            fold = self.sync.fold_for(reduced)
        keep = cover if (eng.fold_sumsq or eng.store_wgrads) else None
        eng.fold_sumsq = eng.store_wgrads = False
        self.arena.step(self.clip, grad_scale=scale, packer=eng, fold=fold, keep=keep)

    def _eager(self, images, masks, edges):
        losses = self._fwd_bwd(images, masks, edges)
        scale = self.sync.finish() if self.sync is not None else 1.0
        self._opt(scale, reduced=self.sync.buckets if self.sync is not None else None)
        return losses

    def _allreduce_flat(self):
        self.sync.reduce_range(0, self.arena.size)

    def _step_split(self, images, masks, edges):
        losses = self._fwd_bwd(images, masks, edges)
        if self.comm:
            self._allreduce_flat()
        self._opt(1.0 / self.world, reduced=[(0, self.arena.size)])
        return losses

    # ---- hand-written backward in segments (multi-GPU graph mode) ---------------------------------------------------
    def _segment_plan(self):
        """[(lo, hi, grad_end_offset)] over trunk blocks, last-to-first; grad_end_offset = arena offset up to which
        gradients are final after the segment (arena order: head, block L-1, ..., block 0, embeddings)."""
        nb = len(self.model.engine.blocks)
        ends = self.arena.unit_ends            # [head, blk L-1, ..., blk 0, embeddings]
        per = (nb + self.n_segments - 1) // self.n_segments
        plan, hi = [], nb
        while hi > 0:
            lo = max(0, hi - per)
            done_units = 1 + (nb - lo)         # head + blocks nb-1 .. lo
            plan.append((lo, hi, ends[done_units - 1] if lo > 0 else ends[-1]))
            hi = lo
        return plan

    def _seg_first(self, images, masks, edges, lo, hi):
        model, eng = self.model, self.model.engine
        ops.begin_zero_pool(images.device)
        self._begin_grads(images)
        feats, tctx = eng.trunk_fwd(images, True, True)
        out, hctx = eng.head_fwd(feats[1:4], True, True)
        leaves = [t.detach().requires_grad_(True) for t in out["predictions"] + [out["edge"]]]
        losses = self.criterion.forward_batched(leaves[:3], leaves[3], masks, edges)
        grads = torch.autograd.grad(losses["loss"], leaves)     # only the fused-loss node: four tiny kernels
        d = eng.head_bwd(hctx, list(grads[:3]), grads[3], None)
        eng.trunk_bwd_begin(tctx, [None] + d)
        eng.trunk_bwd_blocks(lo, hi)
        return {k: v.detach() for k, v in losses.items()}

    def _capture_segmented(self, images, masks, edges):
        model, ar = self.model, self.arena
        eng = model.engine
        self.static = (images.clone(), masks.clone(), edges.clone())
        if ar.m is None:
            ar.m, ar.v = torch.zeros_like(ar.p), torch.zeros_like(ar.p)
        plan = self._segment_plan()
        keep = [ar.p, ar.m, ar.v, ar.step_f] + list(model.buffers())
        snap = [t.clone() for t in keep]

        # CU budget per segment (grid sizes are frozen at capture): the first segment and the optimizer run with no collective in
        # flight -> all CUs; segments 2.. replay while the previous range is being all-reduced -> leave RCCL its CUs.  The warm-up runs
        # under the SAME budgets: the weight-gradient launches -- and with them the set of gradient matrices that are stored whole and kept
        # uncleared from step to step -- depend on the budget, and a warm-up that made other launches than the captured step leaves
        # gradients behind that the captured backward adds to (Arena.step refuses that: the capture failed and every rank fell back to
        # the eager step -- silently but for a log line, found by tools/seg_probe.py)
        comm_cus = COMM_CU_BUDGET if (self.comm or self.force_segmented) else 0

        def run_all():
            (lo, hi, _), rest = plan[0], plan[1:]
            with ops.cu_budget(0):
                losses = self._seg_first(*self.static, lo, hi)
            with ops.cu_budget(comm_cus):
                for (l2, h2, _) in rest:
                    eng.trunk_bwd_blocks(l2, h2)
                eng.trunk_bwd_end()
            if self.comm:
                self._allreduce_flat()
            with ops.cu_budget(0):
                self._opt(1.0 / self.world, reduced=[(0, self.arena.size)])
            return losses

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _i in range(2):
                run_all()
        torch.cuda.current_stream().wait_stream(side)
        for t, c in zip(keep, snap):
            t.copy_(c)
        eng.pack()
        torch.cuda.synchronize()
        if self.sync is not None:
            self.sync.drop_staging()     # the warm-up reduced the arena in one piece: its staging buffer is never used again
        self.comm_stream = torch.cuda.Stream()
        self.segments = []
        g0 = torch.cuda.CUDAGraph()
        with ops.cu_budget(0), torch.cuda.graph(g0, capture_error_mode=graph_capture_mode()):
            self.losses = self._seg_first(*self.static, plan[0][0], plan[0][1])
        self.segments.append(g0)
        pool = g0.pool()
        for k, (lo, hi, _) in enumerate(plan[1:], start=1):
            g = torch.cuda.CUDAGraph()
            with ops.cu_budget(comm_cus), torch.cuda.graph(g, pool=pool, capture_error_mode=graph_capture_mode()):
                eng.trunk_bwd_blocks(lo, hi)
                if k == len(plan) - 1:
                    eng.trunk_bwd_end()
            self.segments.append(g)
        self.graph_b = torch.cuda.CUDAGraph()
        ranges, start = [], 0                  # what _replay_segmented all-reduces, range by range
        for (_, _, end) in plan:
            ranges.append((start, end))
            start = end
        if self.comm:      # the partial-sum buffers and the fold's chunk table (a host-to-device upload) exist BEFORE the capture
            pre = self.sync.fold_for(ranges)
            if pre is not None:
                ar._fold_plan(pre[1])
        with ops.cu_budget(0), torch.cuda.graph(self.graph_b, pool=pool, capture_error_mode=graph_capture_mode()):
            self._opt(1.0 / self.world, reduced=ranges)
        self._plan = plan
        self.graph = g0

    def _replay_segmented(self):
        import torch.distributed as dist
        start = 0
        main = torch.cuda.current_stream()
        for g, (_, _, end) in zip(self.segments, self._plan):
            g.replay()
            ev = torch.cuda.Event()
            ev.record(main)
            self.comm_stream.wait_event(ev)
            if self.comm:
                with torch.cuda.stream(self.comm_stream):
                    self.sync.reduce_range(start, end)     # bf16 payload by default (GradSync)
            start = end
        main.wait_stream(self.comm_stream)
        self.graph_b.replay()

    # ---- capture --------------------------------------------------------------------------------------------
    def _capture(self, images, masks, edges):
        model, ar = self.model, self.arena
        self.static = (images.clone(), masks.clone(), edges.clone())
        if ar.m is None:
            ar.m, ar.v = torch.zeros_like(ar.p), torch.zeros_like(ar.p)
        # warm-up must not advance training: snapshot every piece of state the step mutates, restore after
        keep = [ar.p, ar.m, ar.v, ar.step_f] + list(model.buffers())
        snap = [t.clone() for t in keep]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):        # warm-up on a side stream (allocator + lazy init), as torch requires
            for _i in range(2):
                self._step_split(*self.static)
        torch.cuda.current_stream().wait_stream(side)
        for t, c in zip(keep, snap):
            t.copy_(c)
        model._engine.pack()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if self.world == 1:
            with torch.cuda.graph(self.graph, capture_error_mode=graph_capture_mode()):
                self.losses = self._eager(*self.static)
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=graph_capture_mode()):
                self.losses = self._fwd_bwd(*self.static)
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode=graph_capture_mode()):
                self._opt(1.0 / self.world)

    def __call__(self, images: torch.Tensor, masks: torch.Tensor, edges: torch.Tensor) -> Dict[str, torch.Tensor]:
        model = self.model
        model.train()
        _ = model.engine  # make sure weights are packed before the first step
        if not self.capture:
            return self._eager(images, masks, edges)
        if self.graph is None:
            err = None
            try:
                if self.comm or self.force_segmented:
                    self._capture_segmented(images, masks, edges)
                else:
                    self._capture(images, masks, edges)
            except Exception as e:      # noqa: BLE001 -- re-raised below unless every rank agrees to run eagerly
                err = e
            if self.comm:
                # ranks must take the same path: a rank that fell back to eager launches alone would issue bucketed all-reduces while
                # its peers issue the segment-range ones (a hang, not an error).  Every rank has run the same warm-up collectives by
                # now, so one MIN all-reduce of the outcome is in step on all of them.
                import torch.distributed as dist
                ok = torch.tensor([0 if err is not None else 1], device=images.device, dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok) == 0:
                    logger.warning("hipGraph capture failed on at least one rank (%s): every rank runs the eager step", err)
                    import warnings      # (a log line alone went unnoticed for hours once: the step then runs ~15 % slower, with the all-reduce exposed)
                    warnings.warn(f"hipGraph capture of the multi-GPU train step failed ({err}); every rank runs the eager step", RuntimeWarning)
                    self.capture, self.graph, self.segments = False, None, None
                    return self._eager(images, masks, edges)
            elif err is not None:
                raise err
            if self.graph is not None:
                self._graph_unzeroed = self.arena._unzeroed      # what the captured optimizer leaves uncleared (and its backward stores whole)
        # the captured graphs are frozen to the shapes of the first batch: anything else (a short last batch, another ground-truth size)
        # runs eagerly instead of being broadcast into / rejected by the static buffers
        if any(dst.shape != src.shape for dst, src in zip(self.static, (images, masks, edges))):
            if self.world > 1:
                raise RuntimeError("a batch shape that differs from the captured one needs drop_last=True in multi-GPU graph mode "
                                   "(ranks would disagree on the collective schedule)")
            return self._eager(images, masks, edges)
        for dst, src in zip(self.static, (images, masks, edges)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        # The captured step neither clears the gradient arena at its start nor the matrices its own whole-block launches store (both decided
        # at capture time).  If something else ran since the last replay -- an eager step of another shape, a ragged batch -- and left a
        # different set of matrices uncleared, clear the arena here, outside the graph.
        ar = self.arena
        if not ar._clean or ar._unzeroed != self._graph_unzeroed:
            ar.g.zero_()
        if self.segments is not None:
            self._replay_segmented()
        else:
            self.graph.replay()
        ar._clean, ar._unzeroed = True, self._graph_unzeroed
        return self.losses


def _rank() -> int:
    return torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0


def _world() -> int:
    return torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1


class TrainingMonitor:
    """Running means + best-model bookkeeping (reference engine/trainer.py:42-199, JSON written atomically)."""

    def __init__(self, dir_manager=None):
        self.dir_manager = dir_manager
        self.history: List[Dict] = []
        self.best = -float("inf")
        self._sums: Dict[str, float] = {}
        self._n = 0

    def update_batch(self, metrics: Dict[str, torch.Tensor], n: int):
        for k, v in metrics.items():
            self._sums[k] = self._sums.get(k, 0.0) + float(v) * n
        self._n += n

    def check_best_model(self, metrics: Dict[str, float], key: str = "weighted_f") -> bool:
        """True when `key` improved.  The default is the SAME quantity Trainer.train gates early stopping and the plateau scheduler on
        (weighted F-measure, -loss when validation does not compute it), so "best checkpoint" and "epochs without improvement" cannot
        disagree (the reference tracks weighted_f for both, engine/trainer.py:559-579)."""
        v = float(metrics[key]) if key in metrics else -float(metrics.get("loss", 0.0))
        if v > self.best:
            self.best = v
            return True
        return False

    def all_reduce(self, device):
        """Sums the running totals over the ranks (each rank saw its own shard): every rank then holds the job-wide means, so the
        scheduler / early-stop / best-checkpoint decisions taken from them are identical everywhere."""
        if _world() == 1 or not self._sums:
            return
        keys = sorted(self._sums)
        t = torch.tensor([self._sums[k] for k in keys] + [float(self._n)], dtype=torch.float64, device=device)
        if t.is_cuda and torch.distributed.get_backend() == "gloo":
            t = t.cpu()
        torch.distributed.all_reduce(t)
        t = t.cpu()
        self._sums = {k: float(v) for k, v in zip(keys, t[:-1])}
        self._n = int(round(float(t[-1])))

    def end_epoch(self, epoch: int, phase: str) -> Dict[str, float]:
        m = {k: v / max(self._n, 1) for k, v in self._sums.items()}
        self.history.append({"epoch": epoch, "phase": phase, **m})
        self._sums, self._n = {}, 0
        if self.dir_manager is not None and hasattr(self.dir_manager, "run_dir") and _rank() == 0:   # one writer per job
            os.makedirs(str(self.dir_manager.run_dir), exist_ok=True)
            path = os.path.join(str(self.dir_manager.run_dir), "metrics.json")
            tmp = path + ".tmp"
            with open(tmp, "w") as f:
                json.dump(self.history, f)
            os.replace(tmp, path)
        return m


class Trainer:
    def __init__(self, config: Dict, dir_manager, device: torch.device):
        self.config = config['training']
        self.model_config = dict(config['model'])
        self.device = torch.device(device)
        self.use_amp = self.config.get('use_amp', True)
        self.model_config.setdefault('compute_dtype', 'bf16' if self.use_amp else 'fp32')
        self.model = SPEGNet(self.model_config).to(self.device)
        self.batch_size = self.config['batch_size']
        self.num_epochs = self.config['num_epochs']
        self.grad_clip = self.config.get('gradient_clip', 1.0)
        self.early_stop_patience = self.config.get('early_stop_patience', 15)
        self.save_freq = self.config.get('save_freq', 1)
        self.dir_manager = dir_manager
        self._setup_optimization()
        self.criterion = CODLoss(**self.config['loss']).to(self.device)
        self.monitor = TrainingMonitor(dir_manager)
        self.sync = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # payload type of the gradient all-reduce: the compute dtype unless `training.grad_allreduce_dtype` says otherwise (fp32
            # parity mode must not round its gradients to bf16 on the wire)
            wire = str(self.config.get('grad_allreduce_dtype', self.model_config.get('compute_dtype', 'bf16'))).lower()
            wire = {'bfloat16': 'bf16', 'float32': 'fp32', 'float': 'fp32'}.get(wire, wire)     # (the spellings SPEGNet._DTYPES accepts)
            if wire not in ('bf16', 'fp32'):
                raise ValueError(f"training.grad_allreduce_dtype must be 'bf16' or 'fp32', got {wire!r}")
            self.sync = GradSync(self.arena.g, self.arena.unit_ends, compress_bf16=(wire == 'bf16'))
        self.step_fn = TrainStep(self.model, self.criterion, self.arena, self.grad_clip, self.sync,
                                 capture=bool(self.config.get('capture_graph', False)))
        self._plateau_best, self._plateau_bad = -float("inf"), 0

    # ---- optimisation set-up (reference engine/trainer.py:255-306) --------------------------------------
    def _get_param_groups(self) -> List[Dict]:
        from .arena import group_of
        base_lr = self.config['optimizer']['learning_rate']
        ratio = self.config['optimizer'].get('encoder_lr_ratio', 0.1)
        wd = self.config['optimizer'].get('weight_decay', 0.01)
        groups = [{'params': [], 'weight_decay': 0.0, 'lr': base_lr * ratio}, {'params': [], 'weight_decay': 0.0, 'lr': base_lr * ratio},
                  {'params': [], 'weight_decay': wd, 'lr': base_lr}, {'params': [], 'weight_decay': 0.0, 'lr': base_lr}]
        for name, p in self.model.named_parameters():
            groups[group_of(name)]['params'].append(p)
        return groups

    def _setup_optimization(self):
        opt = self.config['optimizer']
        self.arena = Arena(self.model)
        self.model.mark_params_changed()
        self.arena.set_hyper(opt['learning_rate'], opt.get('weight_decay', 0.01), opt.get('encoder_lr_ratio', 0.1))
        sch = self.config.get('scheduler', {})
        self._sched = dict(factor=sch.get('factor', 0.5), patience=sch.get('patience', 5), min_lr=sch.get('min_lr', 1e-6))

    def scheduler_step(self, metric: float):
        """ReduceLROnPlateau(mode='max') on the device-resident learning rates."""
        if metric > self._plateau_best:
            self._plateau_best, self._plateau_bad = metric, 0
        else:
            self._plateau_bad += 1
            if self._plateau_bad > self._sched['patience']:
                self.arena.scale_lr(self._sched['factor'], self._sched['min_lr'])
                self._plateau_bad = 0

    # ---- the step (reference engine/trainer.py:308-427) --------------------------------------------------
    def _process_batch(self, batch: Dict, is_train: bool) -> Tuple[Dict[str, torch.Tensor], Dict[str, float]]:
        timing = {}
        t0 = time.time()
        images = batch['images'].to(self.device, non_blocking=True)
        masks = [m.to(self.device, non_blocking=True) for m in batch['masks']]
        edges = [e.to(self.device, non_blocking=True) for e in batch['edges']]
        timing['data_time'] = time.time() - t0
        same = all(m.shape == masks[0].shape for m in masks) and all(e.shape == edges[0].shape for e in edges)
        # the fused step (HIP CODLoss) needs square ground truth of ONE size that is a multiple of every prediction size; everything
        # else -- original-resolution masks such as 640x480 or 400x400 against 384 logits -- takes the reference's per-sample path
        S = masks[0].shape[-1]
        fused_ok = (same and masks[0].shape[-2] == S and edges[0].shape == masks[0].shape and images.shape[-1] == images.shape[-2]
                    and all(S % (images.shape[-1] // d) == 0 for d in (8, 4, 2, 1)))
        if self.sync is not None:   # every rank must take the same path (the two paths issue different collectives)
            flag = torch.tensor([1.0 if fused_ok else 0.0], device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            fused_ok = bool(flag.item() > 0.5)
        if is_train and fused_ok:
            t1 = time.time()
            metrics = self.step_fn(images, torch.stack(masks), torch.stack(edges))
            timing['step_time'] = time.time() - t1
        else:
            t1 = time.time()
            self.model.train(is_train)
            with torch.set_grad_enabled(is_train):
                out = self.model(images)
            timing['forward_time'] = time.time() - t1
            preds, eds = [], []
            for i in range(len(masks)):  # per-sample resize to each ground-truth size (ragged batches)
                preds.append([F.interpolate(p[i:i + 1].float(), size=masks[i].shape[-2:], mode='bilinear', align_corners=False)
                              for p in out['predictions']])
                eds.append(F.interpolate(out['edge'][i:i + 1].float(), size=edges[i].shape[-2:], mode='bilinear', align_corners=False))
            metrics = self.criterion(predictions=preds, edge_pred=eds, masks=masks, edges=edges)
            if not is_train:
                # validation metrics on the device (reference engine/trainer.py:482-520 reads weighted_f / s_alpha / mae from them)
                if getattr(self, "_metrics", None) is None:
                    from ..utils.metrics import MetricsProcessor
                    self._metrics = MetricsProcessor()
                cod = self._metrics.compute_metrics([p[-1][0] for p in preds], masks, [e[0] for e in eds], edges)
                metrics.update({k: torch.tensor(v) for k, v in cod.items()})
            if is_train:
                self.arena.zero_grad()
                metrics['loss'].backward()
                scale = self.sync.finish() if self.sync is not None else 1.0
                self.arena.step(self.grad_clip, grad_scale=scale)
                self.model.mark_params_changed()
            metrics = {k: v.detach() for k, v in metrics.items()}
        timing['batch_time'] = time.time() - t0
        return metrics, timing

    def train_epoch(self, loader, epoch: int) -> Dict[str, float]:
        self.model.train()
        for batch in loader:
            metrics, _ = self._process_batch(batch, is_train=True)
            self.monitor.update_batch(metrics, len(batch['masks']))
        self.monitor.all_reduce(self.device)
        return self.monitor.end_epoch(epoch, "train")

    @torch.no_grad()
    def validate(self, loader) -> Dict[str, float]:
        self.model.eval()
        for batch in loader:
            metrics, _ = self._process_batch(batch, is_train=False)
            self.monitor.update_batch(metrics, len(batch['masks']))
        self.monitor.all_reduce(self.device)     # the validation set is sharded over the ranks: the score below is the job-wide mean
        return self.monitor.end_epoch(-1, "val")

    def _loaders(self, dataset_dirs: Sequence[str]):
        """This package's loader (utils/data_loader.py, the reference's surface + pinned memory + optional device preprocessing), or --
        `training.use_host_loader: true` -- the host tree's utils.data_loader when the package is dropped into the reference checkout."""
        kw = dict(dataset_dirs=list(dataset_dirs), model_config=self.model_config, batch_size=self.batch_size,
                  num_workers=self.config.get('num_workers', 8), val_ratio=self.config.get('val_ratio', 0.1))
        world = _world()
        if self.config.get('use_host_loader', False):
            try:
                from utils.data_loader import get_training_loaders as host_loaders   # the reference's module (utils/data_loader.py:214-314)
            except ImportError as e:
                raise ImportError("training.use_host_loader is set but `utils.data_loader` is not importable: run from the reference "
                                  "checkout (its utils/ on sys.path) or unset the flag to use spegnet_amd.utils.data_loader") from e
            if world > 1:
                raise RuntimeError("training.use_host_loader: the reference's loader is single-GPU (no sampler sharding, "
                                   "utils/data_loader.py:287-301); multi-GPU runs need spegnet_amd.utils.data_loader")
            return host_loaders(**kw)
        from ..utils.data_loader import get_training_loaders
        # one shard of the training / validation sets per rank; graph mode (and any multi-GPU run) drops the ragged last batch so that
        # every rank issues the same number of steps and collectives
        return get_training_loaders(device_preprocess=bool(self.config.get('device_preprocess', False)), rank=_rank(), world=world,
                                    drop_last=bool(self.config.get('drop_last', world > 1 or self.step_fn.capture)), **kw)

    def _batches(self, loader):
        """batches one step ahead of the compute stream (pinned H2D + batched device preprocessing on a copy stream)"""
        if self.device.type != 'cuda':
            return loader
        from ..utils.data_loader import DeviceBatcher, prefetch
        ip = self.model_config.get('image_processing', {})
        if getattr(self, "_batcher", None) is None and self.config.get('device_preprocess', False):
            self._batcher = DeviceBatcher(ip.get('target_size', 512), ip.get('normalize_mean', (0.485, 0.456, 0.406)),
                                          ip.get('normalize_std', (0.229, 0.224, 0.225)), self.device)
        return prefetch(loader, getattr(self, "_batcher", None), self.device)

    def train(self, dataset_dirs: Sequence[str]):
        """Training loop of reference engine/trainer.py:522-586: plateau scheduler, early stop (min_delta 1e-4) and best checkpoint are
        driven by the validation weighted F-measure when validation computes it (Evaluator metrics), by -loss otherwise."""
        train_loader, val_loader = self._loaders(dataset_dirs)
        logger.info("Training samples: %d", len(train_loader.dataset))
        resumed = hasattr(self, "_resume_best")
        # (restored by resume(); a fresh run starts below every possible score, so the first validated epoch always becomes the best,
        # also when validation falls back to -loss)
        best, bad = (self._resume_best, getattr(self, "_resume_bad", 0)) if resumed else (-float("inf"), 0)
        min_delta = self.config.get('min_delta', 1e-4)
        start = getattr(self, "_start_epoch", 0)
        for epoch in range(start, self.num_epochs):
            for ld in (train_loader, val_loader):
                if ld is not None and hasattr(getattr(ld, "sampler", None), "set_epoch"):
                    ld.sampler.set_epoch(epoch)       # DistributedSampler: a new permutation per epoch, the same on every rank
            self._epoch_state = (best, bad)
            tr = self.train_epoch(self._batches(train_loader), epoch)
            va = tr
            if val_loader is not None:
                va = self.validate(self._batches(val_loader))
                score = va['weighted_f'] if 'weighted_f' in va else -va['loss']
                self.scheduler_step(score)
                if score - best > min_delta:
                    best, bad = score, 0
                    self._epoch_state = (best, bad)      # (before the save: model_best.pth must carry THIS epoch's early-stop state)
                    if self.monitor.check_best_model(va):
                        self._save_checkpoint(epoch, va, is_best=True)
                else:
                    bad += 1
                self._epoch_state = (best, bad)
                if bad >= self.early_stop_patience:
                    logger.info("Early stopping triggered")
                    break
            logger.info("epoch %d train %s val %s", epoch, tr, va)
            if (epoch + 1) % self.save_freq == 0:
                self._save_checkpoint(epoch, va, is_best=False)

    def resume(self, path: str) -> int:
        """Checkpoint resume (the reference saves but never loads optimizer state, engine/trainer.py:588-606): restores the model,
        the arena's Adam moments / step / learning rates and the scheduler bookkeeping; returns the next epoch."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ckpt['model_state_dict'])
        self.arena.load_state_dict(ckpt['optimizer_state_dict'])
        sch = ckpt.get('scheduler_state_dict') or {}
        self._plateau_best, self._plateau_bad = sch.get('best', self._plateau_best), sch.get('bad', self._plateau_bad)
        # early-stop / best-checkpoint bookkeeping: without it the first validated epoch after a resume would overwrite model_best.pth
        # with a possibly worse model and restart the patience counter
        self._resume_best, self._resume_bad = sch.get('early_stop_best', 0.0), sch.get('early_stop_bad', 0)
        self.monitor.best = sch.get('monitor_best', self.monitor.best)
        self.model.mark_params_changed()
        self._start_epoch = int(ckpt.get('epoch', -1)) + 1
        return self._start_epoch

    def _save_checkpoint(self, epoch: int, metrics: Dict, is_best: bool = False):
        """Same checkpoint schema as reference engine/trainer.py:588-606 (model_state_dict + config are what
        Predictor / main.py read back)."""
        if _rank() != 0:      # ranks hold identical parameters / optimizer state after every step: one writer per job
            return
        es_best, es_bad = getattr(self, "_epoch_state", (0.0, 0))
        ckpt = {'epoch': epoch, 'model_state_dict': {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                'optimizer_state_dict': {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.arena.state_dict().items()},
                'scheduler_state_dict': dict(self._sched, best=self._plateau_best, bad=self._plateau_bad, early_stop_best=es_best,
                                             early_stop_bad=es_bad, monitor_best=self.monitor.best), 'scaler': None,
                'metrics': metrics, 'config': {'training': self.config, 'model': self.model_config}}
        d = str(getattr(self.dir_manager, "checkpoint_dir", getattr(self.dir_manager, "run_dir", ".")))
        os.makedirs(d, exist_ok=True)
        torch.save(ckpt, os.path.join(d, 'model_best.pth' if is_best else f'checkpoint_{epoch:03d}.pth'))
