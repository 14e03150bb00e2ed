"""In-kernel stamps of the halo-tile convolution (dev library, SPG_CONV_HALO_DBG=5): where a phase spends its cycles.
usage: SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so [SPG_CONV_HALO=64|128|256] python tools/halo_stamps.py B H W Ci Co"""
import sys, os, ctypes
os.environ["SPG_CONV_HALO_DBG"] = "5"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from spegnet_amd import ops, _lib
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, W, Ci, device="cuda").to(torch.bfloat16)
wp = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).to(torch.bfloat16)
for _ in range(3):
    ops.gemm_nt(x, wp, conv=(B, H, W, Ci))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 6))()
lib = _lib.load()
assert lib.spg_dev_halo_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 6).astype(np.float64)
n = a[..., 4]
names = ["issue reads + DMA (+epilogue)", "barrier 1 + lgkmcnt(0)", "MFMA cluster", "barrier 2"]
print(f"B{B} {H}x{W} {Ci}->{Co}: phases per wave {n.mean():.0f}; cycles per phase (s_memtime ticks = 100 MHz? see guide; relative shares matter)")
tot = a[..., :4].sum(-1) / n
print(f"  total per phase {tot.mean():.1f}")
for i, nm in enumerate(names):
    v = a[..., i] / n
    print(f"  {nm:32s} {v.mean():8.1f} ({100*v.mean()/tot.mean():4.1f} %)  group0 {v[:, :4].mean():8.1f}  group1 {v[:, 4:].mean():8.1f}")
