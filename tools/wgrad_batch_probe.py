import sys, os
sys.path.insert(0, os.getcwd())
import torch
from spegnet_amd import ops
sys.path.insert(0, "tools")
from conv_check import timeit
dt = torch.bfloat16
for (B, H, W, Ci, Co) in [(1, 384, 384, 64, 64), (2, 384, 384, 64, 64), (4, 384, 384, 64, 64), (8, 384, 384, 64, 64), (16, 384, 384, 64, 64), (4, 384, 384, 128, 64), (8, 384, 384, 128, 64)]:
    x = torch.randn(B, H, W, Ci, device="cuda").to(dt); dy = torch.randn(B, H, W, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, 9 * Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
    t = timeit(lambda: ops.gemm_tn(dy, x, dw, conv=(B, H, W, Ci), dbias=db))
    fl = 2.0 * B * H * W * Co * 9 * Ci
    print(f"B{B} {H}x{W} {Ci}->{Co}: {t*1e6:8.1f} us {fl/t/1e12:6.0f} TF  ({(x.numel()+dy.numel())*2/1e6:.0f} MB operands)", flush=True)
