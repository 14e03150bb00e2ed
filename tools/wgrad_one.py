import sys, os
sys.path.insert(0, os.getcwd())
import torch
from spegnet_amd import ops
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
dt = torch.bfloat16
x = torch.randn(B, H, W, Ci, device="cuda").to(dt); dy = torch.randn(B, H, W, Co, device="cuda").to(dt)
dw = torch.zeros(Co, 9 * Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
for _ in range(6):
    ops.gemm_tn(dy, x, dw, conv=(B, H, W, Ci), dbias=db)
torch.cuda.synchronize()
