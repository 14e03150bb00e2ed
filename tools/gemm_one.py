import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
it = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
out = torch.empty(M, N, device="cuda", dtype=dt)
for _ in range(it):
    ops.gemm_nt(x, w, out=out)
torch.cuda.synchronize()
