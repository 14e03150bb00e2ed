"""Conv3x3 / dense single-problem wgrad (spg_gemm_tn) at the head's shapes: time per launch (hipGraph), dev library A/B via SPG_TN_GROUP_V4."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from nt_check import timeit
mk = lambda *sh: torch.randn(*sh, device="cuda").to(torch.bfloat16)
for B, H, W, Ci, Co in [(8, 96, 96, 128, 128), (8, 48, 48, 192, 64), (8, 192, 192, 64, 64), (8, 96, 96, 64, 128)]:
    x, dy = mk(B, H, W, Ci), mk(B * H * W, Co)
    dw = torch.zeros(Co, 9 * Ci, device="cuda")
    t = timeit(lambda: ops.gemm_tn(dy, x, dw, conv=(B, H, W, Ci)), iters=10)
    print(f"conv wgrad B{B} {H}x{W} {Ci}->{Co}: {t*1e6:7.1f} us {2.0*B*H*W*Co*9*Ci/t/1e12:5.0f} TF", flush=True)
for M, N, K in [(18432, 512, 576), (18432, 512, 288), (4608, 512, 1152), (73728, 144, 152)]:
    x, dy = mk(M, K), mk(M, N)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    t = timeit(lambda: ops.gemm_tn(dy, x, dw, dbias=db), iters=10)
    print(f"dense wgrad {M}x{N}x{K}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:5.0f} TF", flush=True)
