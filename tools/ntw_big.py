import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from spegnet_amd import ops
from ntw_check import timeit
dt = torch.bfloat16
for M, N, K, act, tag in [(36864, 1728, 576, ops.ACT_NONE, "qkv b64"), (36864, 2304, 576, ops.ACT_GELU, "fc1 b64 eval"), (9216, 2304, 576, ops.ACT_GELU_SAVE_GRAD, "fc1 768px"),
                          (9216, 1728, 576, ops.ACT_NONE, "qkv 768px")]:
    x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); b = torch.randn(N, device="cuda")
    pre = torch.empty(M, N, device="cuda", dtype=dt) if act == ops.ACT_GELU_SAVE_GRAD else None
    out = torch.empty(M, N, device="cuda", dtype=dt)
    t = timeit(lambda: ops.gemm_nt(x, w, bias=b, act=act, preact_out=pre, out=out))
    print(f"{tag:14s} {M}x{N}x{K}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.0f} TF", flush=True)
