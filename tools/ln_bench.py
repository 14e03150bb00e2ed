"""Times layernorm fwd / bwd (with and without parameter gradients) at the trunk's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from gemm_bench import timeit

for M, C in [(4608, 576), (18432, 288), (73728, 144), (1152, 1152)]:
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    t0 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    t1 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, None, None, dres=dy))
    t2 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dy))
    mb = M * C * 2 / 1e6
    print(f"[{M:6d},{C:5d}] {mb:6.1f} MB/tensor | fwd {t0*1e6:6.1f} us | bwd dx only {t1*1e6:6.1f} us | bwd + params {t2*1e6:6.1f} us", flush=True)
