"""On-device COD metrics for validation / evaluation, with the surface of the reference's utils/metrics.py
(`MetricsProcessor.compute_metrics(seg_pred, seg_gt, edge_pred, edge_gt)` :142-260 -> {'s_alpha','weighted_f','mae','e_phi','mean_f'
[,'edge_mae','edge_f']}, averaged over the samples).

The reference quantises sigmoid(logits) to uint8 on the GPU, copies every map to the host and runs `py_sod_metrics` in a process pool
(min(42, ncpu-1) workers).  Here the five measures are evaluated where the predictions already are -- tensor arithmetic in float64 on
the device, one sample at a time (ground truth has its original, per-sample size) -- no image-sized transfer crosses PCIe, only a
few scalars per sample (data-dependent branches) and the seven results per batch.  Formulas: S-measure (Fan et al., ICCV 2017: alpha 0.5, object- and region-aware parts), adaptive E-measure (Fan et al., IJCAI
2018, in its counting form), F-measure curve over the 256 uint8 thresholds (beta^2 = 0.3; 'mean_f' is the mean of the curve),
weighted F-measure (Margolin et al., CVPR 2014: exact Euclidean distance transform with nearest-foreground indices, 7x7 Gaussian
sigma 5, beta 1) and MAE, each after the min-max normalisation of the prediction and the `gt > 128` binarisation that
`py_sod_metrics` applies to its uint8 inputs.

PARITY UNPINNED against `py_sod_metrics` itself: the package is not installed in this environment and the reference
holds no metric fixtures.  What IS pinned: tests/test_metrics.py restates the same published formulas with numpy + scipy.ndimage
(distance_transform_edt, convolve) on the CPU and requires agreement to 1e-6 (the distance transform's nearest-pixel choice among
equidistant foreground pixels is the one admissible difference; the test inputs avoid exact ties mattering).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Union

import torch
import torch.nn.functional as F

_EPS = torch.finfo(torch.float64).eps


def _prepare(pred_u8: torch.Tensor, gt_u8: torch.Tensor):
    """uint8 maps -> (pred in [0,1] float64, min-max normalised as py_sod_metrics does; gt bool)"""
    gt = gt_u8 > 128
    pred = pred_u8.to(torch.float64) / 255.0
    lo, hi = pred.min(), pred.max()
    if bool(hi != lo):
        pred = (pred - lo) / (hi - lo)
    return pred, gt


def mae(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return (pred - gt.to(pred.dtype)).abs().mean()


def adaptive_threshold(pred: torch.Tensor) -> torch.Tensor:
    return torch.clamp(2.0 * pred.mean(), max=1.0)


def fmeasure_curve(pred: torch.Tensor, gt: torch.Tensor, beta2: float = 0.3) -> torch.Tensor:
    """F-measure at the 256 thresholds `pred_uint8 >= t` via foreground / background histograms (cumulated from 255 down)."""
    q = (pred * 255).to(torch.uint8).to(torch.int64).flatten()
    g = gt.flatten()
    fg = torch.bincount(q[g], minlength=256).to(torch.float64)
    bg = torch.bincount(q[~g], minlength=256).to(torch.float64)
    tp = torch.cumsum(fg.flip(0), 0)
    ps = tp + torch.cumsum(bg.flip(0), 0)
    T = max(float(g.sum()), 1.0)
    prec = torch.where(ps == 0, torch.ones_like(tp), tp / ps.clamp_min(1.0))
    rec = tp / T
    num = (1 + beta2) * prec * rec
    den = torch.where(num == 0, torch.ones_like(num), beta2 * prec + rec)
    return num / den


def emeasure_adaptive(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """Enhanced-alignment measure at the adaptive threshold, in its counting form: the enhanced matrix takes one of four values (prediction
    fg/bg x ground truth fg/bg), so its sum needs only the four counts."""
    n = float(gt.numel())
    gt_fg = float(gt.sum())
    b = pred >= adaptive_threshold(pred)
    fg_fg = float((b & gt).sum())
    fg_bg = float((b & ~gt).sum())
    fg_ = fg_fg + fg_bg
    bg_ = n - fg_
    if gt_fg == 0:
        s = bg_
    elif gt_fg == n:
        s = fg_
    else:
        bg_fg = gt_fg - fg_fg
        bg_bg = bg_ - bg_fg
        mp, mg = fg_ / n, gt_fg / n
        parts = ((fg_fg, 1 - mp, 1 - mg), (fg_bg, 1 - mp, -mg), (bg_fg, -mp, 1 - mg), (bg_bg, -mp, -mg))
        s = 0.0
        for cnt, a, c in parts:
            align = 2 * (a * c) / (a * a + c * c + _EPS)
            s += (align + 1) ** 2 / 4 * cnt
    return torch.tensor(s / (n - 1 + _EPS), dtype=torch.float64, device=pred.device)


def _s_object(p: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    v = p[g]
    x = v.mean()
    sx = v.std(unbiased=True) if v.numel() > 1 else torch.zeros((), dtype=p.dtype, device=p.device)
    return 2 * x / (x * x + 1 + sx + _EPS)


def _ssim(p: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    n = p.numel()
    x, y = p.mean(), g.mean()
    sx = ((p - x) ** 2).sum() / (n - 1)
    sy = ((g - y) ** 2).sum() / (n - 1)
    sxy = ((p - x) * (g - y)).sum() / (n - 1)
    a = 4 * x * y * sxy
    b = (x * x + y * y) * (sx + sy)
    if bool(a != 0):
        return a / (b + _EPS)
    return torch.ones_like(a) if bool(b == 0) else torch.zeros_like(a)


def smeasure(pred: torch.Tensor, gt: torch.Tensor, alpha: float = 0.5) -> torch.Tensor:
    g = gt.to(pred.dtype)
    y = g.mean()
    if bool(y == 0):
        sm = 1 - pred.mean()
    elif bool(y == 1):
        sm = pred.mean()
    else:
        so = y * _s_object(pred * g, gt) + (1 - y) * _s_object((1 - pred) * (1 - g), ~gt)
        h, w = gt.shape
        idx = torch.nonzero(gt)
        cy = int(torch.round(idx[:, 0].to(torch.float64).mean())) + 1
        cx = int(torch.round(idx[:, 1].to(torch.float64).mean())) + 1
        area = float(h * w)
        sr = 0.0
        for (ys, xs), wt in (((slice(0, cy), slice(0, cx)), cx * cy / area), ((slice(0, cy), slice(cx, w)), cy * (w - cx) / area),
                             ((slice(cy, h), slice(0, cx)), (h - cy) * cx / area), ((slice(cy, h), slice(cx, w)), (h - cy) * (w - cx) / area)):
            if wt > 0:
                sr = sr + wt * _ssim(pred[ys, xs], g[ys, xs])
        sm = alpha * so + (1 - alpha) * sr
    return torch.clamp(sm, min=0.0)


def edt_with_indices(fg: torch.Tensor, chunk: int = 64):
    """Exact Euclidean distance of every pixel to the nearest True pixel of `fg` [H,W], and that pixel's (row, col) -- what
    scipy.ndimage.distance_transform_edt(~fg, return_indices=True) returns.  Separable: nearest foreground row inside each column by
    running maxima / minima (O(HW)), then a minimisation over source columns per row, evaluated for `chunk` rows at a time
    ([chunk, W, W] on the device)."""
    H, W = fg.shape
    dev = fg.device
    big = 10 * (H + W)
    rows = torch.arange(H, device=dev).view(H, 1).expand(H, W)
    up = torch.where(fg, rows, torch.full_like(rows, -big)).cummax(0).values                # nearest fg row at or above
    dn = torch.where(fg, rows, torch.full_like(rows, big)).flip(0).cummin(0).values.flip(0)  # nearest fg row at or below
    use_up = (rows - up) <= (dn - rows)
    nr = torch.where(use_up, up, dn)                                                          # nearest fg row in the same column
    g = (rows - nr).abs().to(torch.float64)
    g = torch.where((nr < 0) | (nr >= H), torch.full_like(g, float(big)), g)
    cols = torch.arange(W, device=dev, dtype=torch.float64)
    dx2 = (cols.view(W, 1) - cols.view(1, W)) ** 2                                           # [x, x']
    dist = torch.empty((H, W), dtype=torch.float64, device=dev)
    src = torch.empty((H, W), dtype=torch.int64, device=dev)
    for y0 in range(0, H, chunk):
        gg = g[y0:y0 + chunk]                                                                 # [c, x']
        cost = dx2.unsqueeze(0) + (gg * gg).unsqueeze(1)                                      # [c, x, x']
        d2, arg = cost.min(2)
        dist[y0:y0 + chunk] = d2.sqrt()
        src[y0:y0 + chunk] = arg
    irow = torch.gather(nr, 1, src)
    return dist, irow.clamp(0, H - 1), src


def _gauss7(sigma: float = 5.0, device=None) -> torch.Tensor:
    ax = torch.arange(-3, 4, dtype=torch.float64, device=device)
    k = torch.exp(-(ax.view(-1, 1) ** 2 + ax.view(1, -1) ** 2) / (2 * sigma * sigma))
    return k / k.sum()


def weighted_fmeasure(pred: torch.Tensor, gt: torch.Tensor, beta2: float = 1.0) -> torch.Tensor:
    if not bool(gt.any()):
        return torch.zeros((), dtype=torch.float64, device=pred.device)
    g = gt.to(torch.float64)
    dst, ir, ic = edt_with_indices(gt)
    E = (pred - g).abs()
    Et = torch.where(gt, E, E[ir, ic])                       # background pixels take the error of their nearest foreground pixel
    EA = F.conv2d(Et[None, None], _gauss7(device=pred.device)[None, None], padding=3)[0, 0]
    mn = torch.where(gt & (EA < E), EA, E)
    B = torch.where(gt, torch.ones_like(dst), 2 - torch.exp(math.log(0.5) / 5 * dst))
    Ew = mn * B
    tpw = g.sum() - Ew[gt].sum()
    fpw = Ew[~gt].sum()
    R = 1 - Ew[gt].mean()
    P = tpw / (tpw + fpw + _EPS)
    return (1 + beta2) * R * P / (R + beta2 * P + _EPS)


def sample_metrics(pred_u8: torch.Tensor, gt_u8: torch.Tensor) -> Dict[str, torch.Tensor]:
    """the five measures of one (prediction, ground truth) pair of uint8 [H,W] maps"""
    pred, gt = _prepare(pred_u8, gt_u8)
    return {"sm": smeasure(pred, gt), "wfm": weighted_fmeasure(pred, gt), "mae": mae(pred, gt), "em": emeasure_adaptive(pred, gt),
            "fm": fmeasure_curve(pred, gt).mean()}


class MetricsProcessor:
    """Drop-in for the reference's MetricsProcessor (utils/metrics.py:90-290): same constructor argument (ignored: there is no process
    pool), same `compute_metrics` contract and result keys."""

    def __init__(self, num_processes: Optional[int] = None):
        self.num_processes = 0

    @torch.no_grad()
    def compute_metrics(self, seg_pred: Union[List[torch.Tensor], torch.Tensor], seg_gt: List[torch.Tensor],
                        edge_pred: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
                        edge_gt: Optional[List[torch.Tensor]] = None) -> Dict[str, float]:
        """seg_pred: LOGITS, list of [1,H,W] or a [B,1,H,W] tensor (already resized to each ground truth); seg_gt: list of {0,1} maps."""
        def quant(p):
            return (p.float().sigmoid() * 255).to(torch.uint8).squeeze()

        def per_sample(preds, gts):
            preds = [quant(p) for p in (preds if isinstance(preds, list) else preds.split(1, 0))]
            out = []
            for p, g in zip(preds, gts):
                g8 = (g.to(p.device).float() * 255).to(torch.uint8).squeeze()
                out.append(sample_metrics(p, g8))
            return out

        seg = per_sample(seg_pred, seg_gt)
        n = len(seg)
        keys = (("s_alpha", "sm"), ("weighted_f", "wfm"), ("mae", "mae"), ("e_phi", "em"), ("mean_f", "fm"))
        vals = [torch.stack([r[k] for r in seg]).mean() for _, k in keys]
        names = [a for a, _ in keys]
        if edge_pred is not None and edge_gt is not None:
            ed = per_sample(edge_pred, edge_gt)
            vals += [torch.stack([r["mae"] for r in ed]).sum() / n, torch.stack([r["fm"] for r in ed]).sum() / n]
            names += ["edge_mae", "edge_f"]
        host = torch.stack(vals).cpu().tolist()          # the only device -> host transfer of the batch
        return dict(zip(names, host))
