"""COD metrics (spegnet_amd/utils/metrics.py) against a numpy + scipy.ndimage restatement of the same published formulas
(S-measure, adaptive E-measure, F-measure curve, weighted F-measure, MAE, with py_sod_metrics' uint8 conventions).  Runs on the CPU
(the implementation is device-agnostic tensor arithmetic).  `py_sod_metrics` itself is absent: parity with it is unpinned."""
import numpy as np
import pytest
import torch
from scipy.ndimage import convolve, distance_transform_edt

from spegnet_amd.utils import metrics as M

EPS = np.spacing(1)


def prep(pred_u8, gt_u8):
    gt = gt_u8 > 128
    pred = pred_u8 / 255.0
    if pred.max() != pred.min():
        pred = (pred - pred.min()) / (pred.max() - pred.min())
    return pred, gt


def np_fm_curve(pred, gt, beta=0.3):
    q = (pred * 255).astype(np.uint8)
    bins = np.linspace(0, 256, 257)
    fg, _ = np.histogram(q[gt], bins=bins)
    bg, _ = np.histogram(q[~gt], bins=bins)
    tp = np.cumsum(np.flip(fg)); ps = tp + np.cumsum(np.flip(bg))
    T = max(np.count_nonzero(gt), 1)
    ps[ps == 0] = 1
    prec = tp / ps
    prec[(tp == 0) & (np.cumsum(np.flip(fg)) + np.cumsum(np.flip(bg)) == 0)] = 1
    rec = tp / T
    num = (1 + beta) * prec * rec
    den = np.where(num == 0, 1, beta * prec + rec)
    return num / den


def np_em(pred, gt):
    thr = min(2 * pred.mean(), 1)
    b = pred >= thr
    n, gfg = gt.size, np.count_nonzero(gt)
    fg_fg, fg_bg = np.count_nonzero(b & gt), np.count_nonzero(b & ~gt)
    fg_, bg_ = fg_fg + fg_bg, n - fg_fg - fg_bg
    if gfg == 0:
        s = bg_
    elif gfg == n:
        s = fg_
    else:
        bg_fg = gfg - fg_fg; bg_bg = bg_ - bg_fg
        mp, mg = fg_ / n, gfg / n
        s = 0
        for cnt, a, c in ((fg_fg, 1 - mp, 1 - mg), (fg_bg, 1 - mp, -mg), (bg_fg, -mp, 1 - mg), (bg_bg, -mp, -mg)):
            s += ((2 * a * c / (a * a + c * c + EPS)) + 1) ** 2 / 4 * cnt
    return s / (n - 1 + EPS)


def np_sm(pred, gt, alpha=0.5):
    g = gt.astype(np.float64)
    y = g.mean()
    if y == 0:
        return max(0, 1 - pred.mean())
    if y == 1:
        return max(0, pred.mean())

    def s_obj(p, m):
        x = p[m].mean(); sx = p[m].std(ddof=1)
        return 2 * x / (x * x + 1 + sx + EPS)

    so = y * s_obj(pred * g, gt) + (1 - y) * s_obj((1 - pred) * (1 - g), ~gt)
    h, w = gt.shape
    rr, cc = np.nonzero(gt)
    cy, cx = int(np.round(rr.mean())) + 1, int(np.round(cc.mean())) + 1

    def ssim(p, q):
        n = p.size; x, yy = p.mean(), q.mean()
        sx = ((p - x) ** 2).sum() / (n - 1); sy = ((q - yy) ** 2).sum() / (n - 1); sxy = ((p - x) * (q - yy)).sum() / (n - 1)
        a = 4 * x * yy * sxy; b = (x * x + yy * yy) * (sx + sy)
        return a / (b + EPS) if a != 0 else (1.0 if b == 0 else 0.0)

    area = h * w
    sr = 0
    for (ys, xs), wt in (((slice(0, cy), slice(0, cx)), cx * cy / area), ((slice(0, cy), slice(cx, w)), cy * (w - cx) / area),
                         ((slice(cy, h), slice(0, cx)), (h - cy) * cx / area), ((slice(cy, h), slice(cx, w)), (h - cy) * (w - cx) / area)):
        if wt > 0:
            sr += wt * ssim(pred[ys, xs], g[ys, xs])
    return max(0, alpha * so + (1 - alpha) * sr)


def np_wfm(pred, gt, beta=1.0):
    if not gt.any():
        return 0.0
    g = gt.astype(np.float64)
    dst, idx = distance_transform_edt(~gt, return_indices=True)
    E = np.abs(pred - g)
    Et = E.copy()
    Et[~gt] = Et[idx[0][~gt], idx[1][~gt]]
    ax = np.arange(-3, 4)
    K = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 50.0); K /= K.sum()
    EA = convolve(Et, K, mode="constant", cval=0)
    mn = np.where(gt & (EA < E), EA, E)
    B = np.where(~gt, 2 - np.exp(np.log(0.5) / 5 * dst), np.ones_like(dst))
    Ew = mn * B
    tpw = g.sum() - Ew[gt].sum(); fpw = Ew[~gt].sum()
    R = 1 - Ew[gt].mean(); P = tpw / (tpw + fpw + EPS)
    return (1 + beta) * R * P / (R + beta * P + EPS)


def blob_case(seed, H=53, W=71, flat_fg=False):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (((yy - H * 0.45) / (H * 0.22)) ** 2 + ((xx - W * 0.55) / (W * 0.3)) ** 2 < 1) | ((yy > H * 0.7) & (xx < W * 0.2))
    pred = np.clip(gt * 0.7 + rng.rand(H, W) * 0.5 - 0.1, 0, 1)
    if flat_fg:
        pred[gt] = 0.8          # constant error on the foreground: the nearest-pixel choice among equidistant pixels cannot matter
    return (pred * 255).astype(np.uint8), (gt * 255).astype(np.uint8)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_metrics_match_numpy_scipy_restatement(seed):
    p8, g8 = blob_case(seed)
    pred, gt = prep(p8.astype(np.float64), g8)
    got = M.sample_metrics(torch.from_numpy(p8), torch.from_numpy(g8))
    assert abs(float(got["mae"]) - np.abs(pred - gt).mean()) < 1e-12
    assert abs(float(got["fm"]) - np_fm_curve(pred, gt).mean()) < 1e-9
    assert abs(float(got["em"]) - np_em(pred, gt)) < 1e-9
    assert abs(float(got["sm"]) - np_sm(pred, gt)) < 1e-9
    assert abs(float(got["wfm"]) - np_wfm(pred, gt)) < 2e-3        # equidistant nearest pixels may be chosen differently from scipy


def test_weighted_f_exact_when_ties_cannot_matter_and_edt_matches_scipy():
    p8, g8 = blob_case(3, flat_fg=True)
    pred, gt = prep(p8.astype(np.float64), g8)
    got = M.sample_metrics(torch.from_numpy(p8), torch.from_numpy(g8))
    assert abs(float(got["wfm"]) - np_wfm(pred, gt)) < 1e-9
    dst, ir, ic = M.edt_with_indices(torch.from_numpy(gt))
    ref = distance_transform_edt(~gt)
    assert np.abs(dst.numpy() - ref).max() < 1e-9
    # the returned index IS a nearest foreground pixel
    yy, xx = np.mgrid[0:gt.shape[0], 0:gt.shape[1]]
    assert gt[ir.numpy(), ic.numpy()].all()
    assert np.abs(np.hypot(yy - ir.numpy(), xx - ic.numpy()) - ref).max() < 1e-9


def test_degenerate_ground_truth_and_processor_keys():
    H, W = 20, 24
    p8 = torch.randint(0, 256, (H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    empty, full = torch.zeros(H, W, dtype=torch.uint8), torch.full((H, W), 255, dtype=torch.uint8)
    a, b = M.sample_metrics(p8, empty), M.sample_metrics(p8, full)
    pe, _ = prep(p8.numpy().astype(np.float64), empty.numpy())
    assert float(a["wfm"]) == 0.0 and abs(float(a["sm"]) - (1 - pe.mean())) < 1e-12 and abs(float(b["sm"]) - pe.mean()) < 1e-12
    mp = M.MetricsProcessor()
    logits = [torch.randn(1, H, W), torch.randn(1, H + 3, W + 5)]
    gts = [(torch.rand(1, H, W) > 0.6).float(), (torch.rand(1, H + 3, W + 5) > 0.6).float()]
    out = mp.compute_metrics(logits, gts, edge_pred=logits, edge_gt=gts)
    assert set(out) == {"s_alpha", "weighted_f", "mae", "e_phi", "mean_f", "edge_mae", "edge_f"}
    assert all(0.0 <= v <= 1.0 for v in out.values())
