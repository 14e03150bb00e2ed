"""What does ONE dependent kernel node of a hipGraph cost on this box when the kernel does (almost) nothing?  A chain of 400 tiny launches of
the product library (spg_cast_bf16 of 8 / 64 k / 2.6 M elements), timed per node."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import _lib
from tools.ped_probe import timeit

for n in (8, 65536, 2654208):
    x = torch.randn(n, device="cuda")
    h = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: _lib.call("spg_cast_bf16", x.data_ptr(), h.data_ptr(), n, 0, st()), iters=400)
    print(f"cast of {n:8d} floats: {us:6.2f} us per node", flush=True)
