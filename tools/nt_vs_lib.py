"""Dev tool: the vendor library's (torch.matmul -> hipBLASLt / rocBLAS) time on the train step's dense NT shapes, beside spg_gemm_nt's.
A yardstick for what these small-M shapes can reach on this chip; the product never calls the library.  usage: python tools/nt_vs_lib.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from tools.nt_check import SHAPES, timeit


def main():
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K, tag in SHAPES[:13]:
        x = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        wt = w.t()
        fl = 2.0 * M * N * K
        t_lib = timeit(lambda: torch.matmul(x, wt, out=out))
        t_spg = timeit(lambda: ops.gemm_nt(x, w, out=out))
        print(f"{tag:9s} {M:6d}x{N:5d}x{K:5d}  lib {t_lib*1e6:7.1f}us {fl/t_lib/1e12:5.0f}TF   spg {t_spg*1e6:7.1f}us {fl/t_spg/1e12:5.0f}TF", flush=True)


if __name__ == "__main__":
    main()
