"""In-kernel stamps of the grouped wgrad kernel (dev library, SPG_TN_GROUP_DEBUG=5): where a 64-row step spends its cycles.
Shares only -- the stamped build's fences forbid overlaps the real kernel has (cdna_hip_programming.md 'In-kernel stamps')."""
import sys, os, ctypes
os.environ["SPG_TN_GROUP_DEBUG"] = "5"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from spegnet_amd import ops, _lib
shapes = [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)]
mk = lambda r, c: torch.randn(r, c, device="cuda").to(torch.bfloat16)
jobs = [(mk(M, N), mk(M, K), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")) for M, N, K in shapes]
for _ in range(3):
    d = []
    ops.gemm_tn_group(jobs, d)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 4))()
lib = _lib.load()
assert lib.spg_dev_tn_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 4).astype(np.float64)
steps = 77.3
v4 = os.environ.get("SPG_TN_GROUP_V4", "1") != "0"
if not v4:
    groups = [("all 8 waves", slice(0, 8), ["top waits (vmcnt)", "barrier", "MFMA block + read drain", "tail (epilogue, commit)"])]
else:
    groups = [("multiplying waves 0-3", slice(0, 4), ["top wait (lgkmcnt)", "barrier", "MFMA block + read drain", "tail (epilogue, commit)"]),
              ("loader waves 4-7", slice(4, 8), ["wait (vmcnt 16)", "barrier", "issue 8 pieces + cursor", "-"])]
for title, sl, names in groups:
    x = a[:, sl, :]
    tot = x.sum(-1)
    print(f"{title}: cycles per step (s_memtime ticks), mean over 256 workgroups; total {tot.mean()/steps:.0f} per step")
    for i, n in enumerate(names):
        print(f"  {n:28s} {x[..., i].mean()/steps:8.0f}   ({100*x[..., i].mean()/tot.mean():4.1f} %)   per wave " +
              " ".join(f"{x[:, w, i].mean()/steps:6.0f}" for w in range(x.shape[1])))
