"""In-kernel stamps of the resident-window attention forward (dev library, SPG_ATTN_STAMPS=1): stage-3 windows of Hiera-L @384, batch 8."""
import sys, os, ctypes
os.environ["SPG_ATTN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from spegnet_amd import ops, _lib
B, H, W, heads, hd, ws = 8, 24, 24, 8, 72, 16
C = heads * hd
qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
bias = torch.randn(3 * C, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attn_fwd(qkv, bias, B, H, W, heads, hd, ws)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2048 * 8 * 4))()
nq = (ctypes.c_int * 2048)()
lib = _lib.load()
assert lib.spg_dev_attn_stamps(buf, nq) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 4).astype(np.float64)[:256]
n = np.frombuffer(nq, dtype=np.int32)[:256]
print("cycles (s_memtime ticks) per workgroup, mean over waves; by window size (queries)")
for q in sorted(set(n.tolist())):
    m = a[n == q]
    print(f"  nq={q:4d} ({len(m):3d} workgroups): setup {m[..., 0].mean():7.0f}  staging+barrier {m[..., 1].mean():7.0f}  query blocks {m[..., 2].mean():7.0f}"
          f"  (max wave {m[..., 2].max(axis=1).mean():7.0f})  total {m[..., 3].mean():7.0f}")
