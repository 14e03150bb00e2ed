"""Micro-benchmark of the C-ABI GEMMs at the shapes of the SPEGNet train step (HIP events, many launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops

SHAPES = [  # (M, N, K, tag)
    (4608, 2304, 576, "fc1"), (4608, 576, 2304, "fc2"), (4608, 1728, 576, "qkv"), (4608, 576, 576, "proj"),
    (4608, 576, 1728, "dqkv"), (18432, 288, 1152, "s2 fc2"), (73728, 144, 576, "s1 fc2"), (73728, 576, 144, "s1 fc1"),
    (1152, 4608, 1152, "s4 fc1"), (18432, 512, 2016, "cfi fuse"),
]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dt = torch.bfloat16
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print(f"{'shape':28s} {'nt plain':>16s} {'nt bias+res':>16s} {'nt gelu+pre':>16s} {'tn':>16s}")
    for M, N, K, tag in SHAPES:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        b = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        pre = torch.empty(M, N, device="cuda", dtype=dt)
        dy = torch.randn(M, N, device="cuda").to(dt)
        dw = torch.zeros(N, K, device="cuda")
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: ops.gemm_nt(x, w, out=out))
        t2 = timeit(lambda: ops.gemm_nt(x, w, bias=b, residual=res, out=out))
        t3 = timeit(lambda: ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU, preact_out=pre, out=out))
        t4 = timeit(lambda: ops.gemm_tn(dy, x, dw))
        f = lambda t: f"{t*1e6:7.1f}us {fl/t/1e12:5.0f}TF"
        print(f"{tag:9s} {M:6d}x{N:5d}x{K:5d} {f(t1):>16s} {f(t2):>16s} {f(t3):>16s} {f(t4):>16s}", flush=True)


if __name__ == "__main__":
    main()
