"""Micro-benchmark of the windowed attention kernels at Hiera-L @384, batch 8 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops

CASES = [  # name, B, H, W, heads, hd, ws, pooled
    ("s3 win16 (x32)", 8, 24, 24, 8, 72, 16, False),
    ("s3 global (x3)", 8, 24, 24, 8, 72, 0, False),
    ("s2 win4 (x5)", 8, 48, 48, 4, 72, 4, False),
    ("s1 win8 (x2)", 8, 96, 96, 2, 72, 8, False),
    ("s4 win8 (x3)", 8, 12, 12, 16, 72, 8, False),
    ("blk44 pooled", 8, 24, 24, 16, 72, 16, True),
    ("blk2 win8 pooled", 8, 96, 96, 4, 72, 8, True),
    ("blk8 win4 pooled", 8, 48, 48, 8, 72, 4, True),
]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dt = torch.bfloat16
for name, B, H, W, heads, hd, ws, pooled in CASES:
    C = heads * hd
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(dt)
    bias = torch.randn(3 * C, device="cuda").to(dt)
    qp = idx = None
    if pooled:
        qp, idx = ops.maxpool2_fwd(qkv, B, H, W, C, 3 * C, 0)
    out, lse = ops.attn_fwd(qkv, bias, B, H, W, heads, hd, ws, q_pooled=qp)
    dout = torch.randn_like(out)
    db = torch.zeros(3 * C, device="cuda")
    tf = timeit(lambda: ops.attn_fwd(qkv, bias, B, H, W, heads, hd, ws, q_pooled=qp))
    tb = timeit(lambda: ops.attn_bwd(qkv, bias, out, dout, lse, db, B, H, W, heads, hd, ws, q_pooled=qp))
    print(f"{name:16s} fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
