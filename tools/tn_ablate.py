"""Grouped wgrad kernel, stage-3 block: full kernel vs ablations (SPG_TN_GROUP_DEBUG = 2 no MFMAs, 3 no fragment reads, 4 no fill)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from nt_check import timeit
shapes = [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)]
mk = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16)
jobs = [(mk(M, N), mk(M, K), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")) for M, N, K in shapes]
defer = []
def run():
    ops.gemm_tn_group(jobs, defer)
    defer.clear()
t = timeit(run, iters=10)
print(f"SPG_TN_GROUP_DEBUG={os.environ.get('SPG_TN_GROUP_DEBUG','0')}: {t*1e6:7.1f} us")
