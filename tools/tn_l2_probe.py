"""Is the grouped wgrad kernel bound by L2 misses (HBM / fabric) or by the CU-side fill path?  8 jobs of 4608 x 576 x 576 with DISTINCT
operands (72 operand panels per 64-row step) against the same 8 jobs all reading ONE dY and ONE X (9 panels per step: everything
after the first touch is an L2 hit).  Same tiles, same steps, same LDS / MFMA work."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from nt_check import timeit

M, N, K, J = 4608, 576, 576, 8
mk = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16)
dist = [(mk(M, N), mk(M, K), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")) for _ in range(J)]
dy0, x0 = mk(M, N), mk(M, K)
same = [(dy0, x0, torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")) for _ in range(J)]
fl = 2.0 * M * N * K * J
for name, jobs in (("distinct operands", dist), ("shared operands (L2-resident)", same)):
    t = timeit(lambda: ops.gemm_tn_group(jobs), iters=10)
    print(f"{name:32s} {t*1e6:7.1f} us  {fl/t/1e12:5.0f} TF", flush=True)
# and the real stage-3 set for reference
shapes = [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)]
jobs = [(mk(M_, N_), mk(M_, K_), torch.zeros(N_, K_, device="cuda"), torch.zeros(N_, device="cuda")) for M_, N_, K_ in shapes]
t = timeit(lambda: ops.gemm_tn_group(jobs), iters=10)
print(f"{'stage-3 block':32s} {t*1e6:7.1f} us  {sum(2.0*a*b*c for a,b,c in shapes)/t/1e12:5.0f} TF")
