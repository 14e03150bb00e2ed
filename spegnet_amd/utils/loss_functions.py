"""CODLoss for the MI355X path (reference: utils/loss_functions.py:37-295, resize loop engine/trainer.py:358-383).

Same constructor and the same `forward(predictions[B][scales], edge_pred[B], masks, edges)` contract as the
reference, plus `forward_batched`, which evaluates the identical formula for a whole batch at once when every
mask has the same size (the synthetic / fixed-resolution training case) instead of B x ~70 tiny launches.
Round-1 status: evaluated with stock torch tensor ops on the device (SURVEY.md §8(f) rank 1 keeps the fused HIP
loss kernel as the next row); it is not part of the C ABI yet.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        """predictions: 3 x [B,1,h,w] logits; edge [B,1,h,w]; masks, edges [B,1,S,S] (all the same size)."""
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
