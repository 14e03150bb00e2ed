"""Correctness + timing of spg_gemm_nt over the train step's shapes and epilogues (dev library: SPG_NT_V3=0/1 selects the kernel family).
usage: SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so [SPG_NT_V3=0] python tools/nt_check.py [time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spegnet_amd import ops

SHAPES = [(4608, 2304, 576, "fc1"), (4608, 576, 2304, "fc2"), (4608, 1728, 576, "qkv"), (4608, 576, 576, "proj"), (4608, 576, 1728, "dqkv"),
          (18432, 288, 1152, "s2 fc2"), (18432, 1152, 288, "s2 fc1"), (73728, 144, 576, "s1 fc2"), (73728, 576, 144, "s1 fc1"), (73728, 432, 144, "s1 qkv"),
          (1152, 4608, 1152, "s4 fc1"), (1152, 1152, 4608, "s4 fc2"), (18432, 512, 2016, "cfi fuse"), (300, 200, 144, "ragged"), (77, 72, 136, "tiny"),
          (129, 64, 64, "1 step"), (1000, 136, 200, "k tail")]


def timeit(fn, iters=20):
    """seconds per launch, from a hipGraph of `iters` back-to-back launches (no CPU launch overhead in the number)"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        from spegnet_amd import ops as o
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e-3


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20))


def main():
    dt = torch.bfloat16
    do_time = len(sys.argv) > 1 and sys.argv[1] == "time"
    g = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    for M, N, K, tag in SHAPES:
        x = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        b = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g).to(dt)
        h = torch.randn(M, N, device="cuda", generator=g).to(dt)
        ref = x.float() @ w.float().t()
        out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        pre = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        errs = []
        ops.gemm_nt(x, w, out=out); errs.append(rel(out, ref))
        out.fill_(float("nan")); ops.gemm_nt(x, w, bias=b, residual=res, out=out); errs.append(rel(out, ref + b + res.float()))
        out.fill_(float("nan")); ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU, preact_out=pre, out=out)
        errs.append(max(rel(out, F.gelu(ref + b)), rel(pre, ref + b)))
        out.fill_(float("nan")); ops.gemm_nt(x, w, gelu_h=h, out=out)
        hf = h.float(); hg = torch.autograd.functional.vjp(F.gelu, hf, torch.ones_like(hf))[1]
        errs.append(rel(out, ref * hg))
        worst = max(worst, max(errs))
        line = f"{tag:9s} {M:6d}x{N:5d}x{K:5d} err " + " ".join(f"{e:.1e}" for e in errs)
        if do_time:
            fl = 2.0 * M * N * K
            t1 = timeit(lambda: ops.gemm_nt(x, w, out=out))
            t2 = timeit(lambda: ops.gemm_nt(x, w, bias=b, residual=res, out=out))
            t3 = timeit(lambda: ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU, preact_out=pre, out=out))
            f = lambda t: f"{t*1e6:7.1f}us {fl/t/1e12:5.0f}TF"
            line += f" | plain {f(t1)} bias+res {f(t2)} gelu+pre {f(t3)}"
        print(line, flush=True)
    # conv3x3 (implicit GEMM)
    for B, H, W_, Ci, Co in [(8, 96, 96, 128, 128), (8, 48, 48, 192, 64), (2, 24, 24, 64, 128), (8, 192, 192, 64, 64)]:
        x = torch.randn(B, H, W_, Ci, device="cuda", generator=g).to(dt)
        w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (9 * Ci) ** -0.5)
        wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(dt)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wp.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        out = ops.gemm_nt(x, wp, conv=(B, H, W_, Ci))
        e = rel(out.view(-1, Co), ref)
        worst = max(worst, e)
        line = f"conv3x3 B{B} {H}x{W_} {Ci}->{Co} err {e:.1e}"
        if do_time:
            t = timeit(lambda: ops.gemm_nt(x, wp, conv=(B, H, W_, Ci)))
            line += f" | {t*1e6:7.1f}us {2.0*B*H*W_*Co*9*Ci/t/1e12:5.0f}TF"
        print(line, flush=True)
    print("worst", worst)
    assert worst < 2.5e-2, worst


if __name__ == "__main__":
    main()
