"""Correctness + timing of the 256-row variable-width dense NT kernel (csrc/nt_wide.hip) through spg_gemm_nt on the trunk's wide shapes.
usage: SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so SPG_NT_WIDE=0|1 python tools/ntw_check.py [time]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spegnet_amd import ops

# (M, N, K, act, tag)   act: none | gelu | gelu_d | mulh
SHAPES = [(4608, 2304, 576, "gelu_d", "s3 fc1"), (4608, 2304, 576, "mulh", "s3 dfc2"), (4608, 1728, 576, "none", "s3 qkv"),
          (4608, 2304, 576, "gelu", "s3 fc1 eval"), (4500, 1728, 576, "none", "ragged M"), (4608, 2304, 576, "none", "nobias"),
          (18432, 1152, 288, "gelu_d", "s2 fc1"), (18432, 1152, 288, "mulh", "s2 dfc2"), (18432, 864, 288, "none", "s2 qkv"),
          (18432, 512, 576, "none", "cfi s2"), (5120, 1600, 512, "gelu_d", "other")]
if os.environ.get("NTW_ONLY"):
    SHAPES = [s_ for s_ in SHAPES if any(o in s_[4] for o in os.environ["NTW_ONLY"].split(","))]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


def gelu_grad(x):
    return 0.5 * (1 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def main():
    dt = torch.bfloat16
    do_time = "time" in sys.argv
    g = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    tot = 0.0
    for M, N, K, act, tag in SHAPES:
        x = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        bias = None if tag == "nobias" or act == "mulh" else torch.randn(N, device="cuda", generator=g)
        h = torch.randn(M, N, device="cuda", generator=g).to(dt) if act == "mulh" else None
        out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        out2 = torch.full((M, N), float("nan"), device="cuda", dtype=dt) if act in ("gelu", "gelu_d") else None
        code = {"none": ops.ACT_NONE, "gelu": ops.ACT_GELU, "gelu_d": ops.ACT_GELU_SAVE_GRAD, "mulh": ops.ACT_MUL_H}[act]
        run = lambda: ops.gemm_nt(x, w, bias=bias, act=code, gelu_h=h, out=out, preact_out=out2)
        run()
        torch.cuda.synchronize()
        pre = x.float() @ w.float().t() + (bias if bias is not None else 0.0)
        errs = []
        if act == "none":
            errs.append(("C", out, pre))
        elif act == "gelu":
            errs += [("C", out, F.gelu(pre)), ("C2", out2, pre)]
        elif act == "gelu_d":
            errs += [("C", out, F.gelu(pre)), ("C2", out2, gelu_grad(pre))]
        else:
            errs.append(("C", out, pre * h.float()))
        line = f"{tag:12s} {M}x{N}x{K} {act:6s}"
        for nm, got, ref in errs:
            bad = ~torch.isfinite(got.float())
            e = float((got.float() - ref).abs().max() / ref.abs().max()) if not bad.any() else float("inf")
            worst = max(worst, e)
            line += f" {nm} err {e:.2e}" + (f" NONFINITE {int(bad.sum())}" if bad.any() else "")
            if e > 2e-2:
                d = (got.float() - ref).abs()
                idx = (d > 2e-2 * ref.abs().max()).nonzero()
                line += f" | {len(idx)} bad, first {idx[:3].tolist()} last {idx[-1:].tolist()}"
        if do_time:
            t = timeit(run)
            tot += t
            line += f" | {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF"
        print(line, flush=True)
    if tot:
        print(f"sum {tot * 1e6:.1f} us")
    print("worst", worst)
    assert worst < 2e-2, worst


if __name__ == "__main__":
    main()
