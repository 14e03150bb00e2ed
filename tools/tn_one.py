import sys, os
sys.path.insert(0, os.getcwd())
import torch
from spegnet_amd import ops
shapes = [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)]
jobs = []
for M, N, K in shapes:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    jobs.append((dy, x, torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")))
for _ in range(6):
    ops.gemm_tn_group(jobs)
torch.cuda.synchronize()
