"""In-kernel stamps of the wide dense NT kernel (dev library, SPG_NT_WIDE_DBG=5): where a phase spends its cycles.
usage: SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so python tools/ntw_stamps.py M N K [none|gelu_d|mulh]"""
import sys, os, ctypes
os.environ["SPG_NT_WIDE_DBG"] = "5"
os.environ["SPG_NT_WIDE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from spegnet_amd import ops, _lib
M, N, K = [int(v) for v in sys.argv[1:4]]
act = sys.argv[4] if len(sys.argv) > 4 else "none"
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
h = torch.randn(M, N, device="cuda").to(torch.bfloat16) if act == "mulh" else None
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if act == "gelu_d" else None
code = {"none": ops.ACT_NONE, "gelu_d": ops.ACT_GELU_SAVE_GRAD, "mulh": ops.ACT_MUL_H}[act]
for _ in range(3):
    ops.gemm_nt(x, w, act=code, gelu_h=h, out=out, preact_out=out2)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 6))()
lib = _lib.load()
assert lib.spg_dev_ntw_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 6).astype(np.float64)
a = a[a[..., 4].sum(-1) > 0]
n = a[..., 4]
names = ["issue reads + DMA (+epilogue)", "barrier 1 + lgkmcnt(0)", "MFMA cluster", "barrier 2"]
print(f"{M}x{N}x{K} {act}: {a.shape[0]} workgroups, phases per wave {n.mean():.0f}; s_memtime ticks (100 MHz: 1 tick = 10 ns)")
tot = a[..., :4].sum(-1) / n
print(f"  total per phase {tot.mean():.2f} ticks; whole kernel per workgroup {a[..., 5].mean():.0f} ticks (max {a[..., 5].max():.0f}), in phases {a[..., :4].sum(-1).mean():.0f}")
for i, nm in enumerate(names):
    v = a[..., i] / n
    print(f"  {nm:32s} {v.mean():8.2f} ({100*v.mean()/tot.mean():4.1f} %)  group0 {v[:, :4].mean():8.2f}  group1 {v[:, 4:].mean():8.2f}")
