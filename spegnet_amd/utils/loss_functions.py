"""CODLoss for the MI355X path (reference: utils/loss_functions.py:37-295, resize loop engine/trainer.py:358-383).

Same constructor and the same `forward(predictions[B][scales], edge_pred[B], masks, edges)` contract as the
reference, plus `forward_batched`, which evaluates the identical formula for a whole batch at once when every
mask has the same size (the synthetic / fixed-resolution training case) instead of B x ~70 tiny launches.
`forward_batched` runs on the fused HIP kernels of csrc/loss.hip (weight map, per-scale reductions, analytic
gradients).  The per-sample `forward` (ragged ground-truth sizes, the reference's contract) and
`forward_batched_torch` (cross-check) use stock torch tensor ops.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# one launch for the four maps' reductions and two for their gradients (csrc/loss.hip: *_all) instead of four and seven; the per-map
# entry points stay (tests compare the two bit for bit)
BATCHED_LAUNCHES = os.environ.get("SPG_LOSS_BATCHED", "1") != "0"


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _ints(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


class _FusedCODLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, p3, edge, masks, edges, cfg):
        from .. import _lib
        sw, bw, bce_w, iou_w, edge_w, alpha, gamma = cfg
        B, S = masks.shape[0], masks.shape[-1]
        assert masks.shape[-2] == S and edges.shape == masks.shape, "fused CODLoss needs square, equal-size ground truth"
        if not masks.is_cuda:
            raise RuntimeError("fused CODLoss runs only on an MI355X (HIP) device")
        dev = masks.device
        masks = masks.float().contiguous(); edges = edges.float().contiguous()
        preds = [p.contiguous() for p in (p1, p2, p3, edge)]
        dt = _lib.SPG_BF16 if preds[0].dtype == torch.bfloat16 else _lib.SPG_F32
        assert all(p.dtype == preds[0].dtype for p in preds)
        st = torch.cuda.current_stream().cuda_stream
        from .. import ops
        # every sum below is a deterministic reduction (partials + fixed-order finish): nothing needs zeroing
        buf = torch.empty(B * 4 + 4 * B * 3 + 3, dtype=torch.float32, device=dev)
        stats, seg_sums, edge_sums, out = buf[:B * 4], buf[B * 4:B * 4 + 9 * B], buf[B * 13:B * 16], buf[B * 16:]
        wmap = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        lib = _lib.load()
        nws = max(lib.spg_loss_workspace_floats(B, S), lib.spg_loss_reduce_all_workspace_floats(B))
        ws = ops.red_scratch(dev, nws)   # the launches are stream-ordered: one scratch serves them all
        _lib.call("spg_loss_weight_map", masks.data_ptr(), edges.data_ptr(), wmap.data_ptr(), stats.data_ptr(), B, S, float(bw),
                  ws.data_ptr(), nws, ops.red_counters(dev, B), st)
        if BATCHED_LAUNCHES:             # the four maps' reductions side by side in one launch (same blocks per map: same sums)
            _lib.call("spg_loss_reduce_all", dt, _ptrs(preds), _ints([p.shape[-2] for p in preds]), _ints([p.shape[-1] for p in preds]),
                      masks.data_ptr(), edges.data_ptr(), wmap.data_ptr(), stats.data_ptr(), seg_sums.data_ptr(), edge_sums.data_ptr(), B, S,
                      float(alpha), float(gamma), ws.data_ptr(), nws, ops.red_counters(dev, 4 * B), st)
        else:
            for i in range(3):
                h, w = preds[i].shape[-2:]
                _lib.call("spg_loss_reduce", dt, preds[i].data_ptr(), masks.data_ptr(), wmap.data_ptr(), stats.data_ptr(),
                          seg_sums[i * 3 * B:].data_ptr(), B, S, h, w, 0, 0.0, 0.0, ws.data_ptr(), nws, ops.red_counters(dev, B), st)
            h, w = preds[3].shape[-2:]
            _lib.call("spg_loss_reduce", dt, preds[3].data_ptr(), edges.data_ptr(), None, stats.data_ptr(), edge_sums.data_ptr(), B, S, h, w,
                      1, float(alpha), float(gamma), ws.data_ptr(), nws, ops.red_counters(dev, B), st)
        _lib.call("spg_loss_finalize", stats.data_ptr(), seg_sums.data_ptr(), edge_sums.data_ptr(), out.data_ptr(), B, S, float(sw[0]),
                  float(sw[1]), float(sw[2]), float(bce_w), float(iou_w), float(edge_w), st)
        ctx.saved = (preds, masks, edges, wmap, stats, seg_sums, edge_sums, cfg, dt)
        ctx.set_materialize_grads(False)     # (seg_loss / edge_loss are reported, not differentiated: no zero-filled gradients for them)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_seg, g_edge):
        from .. import _lib
        preds, masks, edges, wmap, stats, seg_sums, edge_sums, cfg, dt = ctx.saved
        sw, bw, bce_w, iou_w, edge_w, alpha, gamma = cfg
        B, S = masks.shape[0], masks.shape[-1]
        st = torch.cuda.current_stream().cuda_stream
        go = g_loss.float().contiguous() if g_loss is not None else None
        if g_loss is None:
            return (None,) * 7
        if BATCHED_LAUNCHES:             # two launches: dL/dz of every full-res pixel of the four maps, then the bilinear adjoints
            dz = torch.empty((4, B, S, S), dtype=torch.float32, device=masks.device)
            grads = [torch.empty_like(p) for p in preds]
            coefs = (ctypes.c_float * 4)(float(sw[0]) / B, float(sw[1]) / B, float(sw[2]) / B, float(edge_w) / B)
            _lib.call("spg_loss_grad_all", dt, _ptrs(preds), _ptrs(grads), _ints([p.shape[-2] for p in preds]), _ints([p.shape[-1] for p in preds]),
                      coefs, masks.data_ptr(), edges.data_ptr(), wmap.data_ptr(), stats.data_ptr(), seg_sums.data_ptr(), edge_sums.data_ptr(),
                      go.data_ptr(), B, S, float(bce_w), float(iou_w), float(alpha), float(gamma), dz.data_ptr(), st)
            return grads[0], grads[1], grads[2], grads[3], None, None, None
        dz = torch.empty((B, S, S), dtype=torch.float32, device=masks.device)   # scratch of the two-pass gradient (stream-ordered reuse)
        grads = []
        for i in range(3):
            h, w = preds[i].shape[-2:]
            d = torch.empty_like(preds[i])
            _lib.call("spg_loss_grad", dt, preds[i].data_ptr(), masks.data_ptr(), wmap.data_ptr(), stats.data_ptr(),
                      seg_sums[i * 3 * B:].data_ptr(), go.data_ptr() if go is not None else None, d.data_ptr(), B, S, h, w, 0,
                      float(sw[i]) / B, float(bce_w), float(iou_w), 0.0, 0.0, dz.data_ptr(), st)
            grads.append(d)
        h, w = preds[3].shape[-2:]
        d = torch.empty_like(preds[3])
        _lib.call("spg_loss_grad", dt, preds[3].data_ptr(), edges.data_ptr(), None, stats.data_ptr(), edge_sums.data_ptr(),
                  go.data_ptr() if go is not None else None, d.data_ptr(), B, S, h, w, 1, float(edge_w) / B, 0.0, 0.0, float(alpha),
                  float(gamma), dz.data_ptr(), st)
        grads.append(d)
        return grads[0], grads[1], grads[2], grads[3], None, None, None


class CODLoss(nn.Module):
    def __init__(self, scale_weights: Optional[List[float]] = None, boundary_weight: float = 5.0, bce_weight: float = 0.4,
                 iou_weight: float = 0.6, edge_weight: float = 0.75, edge_focal_alpha: float = 0.75,
                 edge_focal_gamma: float = 2.0):
        super().__init__()
        self.scale_weights = list(scale_weights or [0.2, 0.3, 0.5])
        self.boundary_weight = boundary_weight
        self.bce_weight = bce_weight
        self.iou_weight = iou_weight
        self.edge_weight = edge_weight
        self.edge_focal_alpha = edge_focal_alpha
        self.edge_focal_gamma = edge_focal_gamma
        k = torch.tensor([[-1., -1., -1.], [-1., 8., -1.], [-1., -1., -1.]]).view(1, 1, 3, 3)
        self.register_buffer('boundary_kernel', k)

    # masks [N,1,H,W] -> weight maps [N,1,H,W]
    def _weights(self, m: torch.Tensor) -> torch.Tensor:
        lap = F.conv2d(m, self.boundary_kernel.to(m), padding=1).abs()
        dist = (F.avg_pool2d(m, 31, 1, 15) - m).abs()
        return 1.0 + self.boundary_weight * (lap + dist)

    @staticmethod
    def _pos_weight(t: torch.Tensor) -> torch.Tensor:
        npos = t.sum((2, 3), keepdim=True)
        nneg = (1 - t).sum((2, 3), keepdim=True)
        return (nneg / (npos + 1e-7)).clamp(0.1, 10.0)

    def _structure(self, pred, m, w, pw):
        bce = F.binary_cross_entropy_with_logits(pred, m, pos_weight=pw, reduction='none')
        wbce = (w * bce).sum((2, 3)) / w.sum((2, 3))
        s = torch.sigmoid(pred)
        inter = (s * m * w).sum((2, 3))
        union = ((s + m) * w).sum((2, 3))
        wiou = 1 - (inter + 1) / (union - inter + 1)
        return self.bce_weight * wbce + self.iou_weight * wiou          # [N,1]

    def _edge(self, pred, t):
        s = torch.sigmoid(pred)
        pw = self._pos_weight(t)
        pt = t * s + (1 - t) * (1 - s)
        focal = -pw * self.edge_focal_alpha * (1 - pt).pow(self.edge_focal_gamma) * torch.log(pt.clamp(min=1e-7))
        inter = (s * t).sum((2, 3))
        union = s.sum((2, 3)) + t.sum((2, 3))
        dice = 1 - (2 * inter + 1) / (union + 1)
        return focal.mean((1, 2, 3)) + dice.mean(1)                      # [N]

    def forward_batched(self, predictions: Sequence[torch.Tensor], edge: torch.Tensor, masks: torch.Tensor,
                        edges: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Fused HIP CODLoss.  predictions: 3 x [B,1,h,w] logits; edge [B,1,h,w]; masks, edges [B,1,S,S] (one size,
        square, a multiple of every prediction size).  Gradients reach the predictions through `loss` only."""
        cfg = (tuple(self.scale_weights), self.boundary_weight, self.bce_weight, self.iou_weight, self.edge_weight,
               self.edge_focal_alpha, self.edge_focal_gamma)
        loss, seg, edge_l = _FusedCODLoss.apply(predictions[0], predictions[1], predictions[2], edge, masks, edges, cfg)
        return {'loss': loss, 'seg_loss': seg, 'edge_loss': edge_l}

    def forward_batched_torch(self, predictions: Sequence[torch.Tensor], edge: torch.Tensor, masks: torch.Tensor,
                              edges: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Same formula with stock torch ops (test cross-check)."""
        size = masks.shape[-2:]
        masks, edges = masks.float(), edges.float()
        w = self._weights(masks)
        pw = self._pos_weight(masks)
        seg = 0.0
        for p, sw in zip(predictions, self.scale_weights):
            p = p.float()
            if p.shape[-2:] != size:
                p = F.interpolate(p, size=size, mode='bilinear', align_corners=False)
            seg = seg + sw * self._structure(p, masks, w, pw).mean()
        e = F.interpolate(edge.float(), size=edges.shape[-2:], mode='bilinear', align_corners=False)
        edge_l = self._edge(e, edges).mean()
        return {'loss': seg + self.edge_weight * edge_l, 'seg_loss': seg, 'edge_loss': edge_l}

    def forward(self, predictions: List[List[torch.Tensor]], edge_pred: List[torch.Tensor], masks: List[torch.Tensor],
                edges: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Reference contract: predictions[b][scale] and edge_pred[b] already resized to masks[b] / edges[b]."""
        B = len(masks)
        seg_t, edge_t = 0.0, 0.0
        for i in range(B):
            m = masks[i].unsqueeze(0).float()
            w, pw = self._weights(m), self._pos_weight(m)
            seg = 0.0
            for p, sw in zip(predictions[i], self.scale_weights):
                seg = seg + sw * self._structure(p.float(), m, w, pw).mean()
            edge_t = edge_t + self._edge(edge_pred[i].float(), edges[i].unsqueeze(0).float()).mean()
            seg_t = seg_t + seg
        seg_a, edge_a = seg_t / B, edge_t / B
        return {'loss': seg_a + self.edge_weight * edge_a, 'seg_loss': seg_a, 'edge_loss': edge_a}
