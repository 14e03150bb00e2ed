"""Grouped stream-K wgrad vs one gemm_tn per layer, for the wgrad sets of the trunk blocks (batch 8 @384)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from gemm_bench import timeit

SETS = {
    "stage 3 block (x36)": [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)],
    "stage 2 block (x6)": [(18432, 288, 1152), (18432, 1152, 288), (18432, 288, 288), (18432, 864, 288)],
    "stage 1 block (x2)": [(73728, 144, 576), (73728, 576, 144), (73728, 144, 144), (73728, 432, 144)],
    "stage 4 block (x4)": [(1152, 1152, 4608), (1152, 4608, 1152), (1152, 1152, 1152), (1152, 3456, 1152)],
}
for name, shapes in SETS.items():
    jobs = []
    for i, (M, N, K) in enumerate(shapes):
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        jobs.append((dy, x, torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")))
    fl = sum(2.0 * M * N * K for M, N, K in shapes)
    t_sep = timeit(lambda: [ops.gemm_tn(dy, x, dw, dbias=db) for dy, x, dw, db in jobs])
    t_grp = timeit(lambda: ops.gemm_tn_group(jobs))
    print(f"{name:22s} separate {t_sep*1e6:7.1f} us ({fl/t_sep/1e12:4.0f} TF)   grouped {t_grp*1e6:7.1f} us ({fl/t_grp/1e12:4.0f} TF)", flush=True)
