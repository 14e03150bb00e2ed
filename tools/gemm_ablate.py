"""Times gemm_nt (plain epilogue) at trunk shapes; run under SPG_GEMM_DEBUG=0/1/2 to split fill time from MFMA+LDS time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops
from gemm_bench import timeit

for M, N, K, tag in [(4608, 2304, 576, "fc1"), (4608, 576, 2304, "fc2"), (4608, 1728, 576, "qkv"), (18432, 1152, 288, "s2 fc1"),
                     (8192, 4096, 4096, "big")]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    if os.environ.get("ZERO") == "1":   # zero operands: separates the clock the chip holds on random data from the kernel's cycles
        x.zero_(); w.zero_()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm_nt(x, w, out=out))
    steps = -(-M // 128) * -(-N // 128) * (K // 64)
    print(f"{tag:8s} {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:6.0f} TF   {t*1e6/ -(-steps // 256):6.3f} us/step-round", flush=True)
