"""Whole-block wgrads of three stage-3 trunk blocks in one launch (ops.gemm_tn_blocks) against the same work as three grouped tile
launches + their batched slab reduce (ops.gemm_tn_group with deferral), hipGraph-timed so host launch cost is out of the picture.
Operands of the three trunk blocks are distinct buffers (cold-ish panels, as in the step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
NBLK = int(sys.argv[2]) if len(sys.argv) > 2 else 3
LAYER = [(1728, 576), (576, 576), (2304, 576), (576, 2304)]
if len(sys.argv) > 3 and sys.argv[3] == "stage4":
    LAYER = [(3456, 1152), (1152, 1152), (4608, 1152), (1152, 4608)]


def graph_time(fn, reps=20, iters=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * iters) * 1e-3


blocks = []
for b in range(NBLK):
    jobs = []
    for N, K in LAYER:
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        jobs.append((dy, x, torch.zeros(N, K, device="cuda"), None if os.environ.get("NOBIAS") else torch.zeros(N, device="cuda")))
    blocks.append(jobs)
flat = [j for jobs in blocks for j in jobs]
fl = sum(2.0 * M * j[0].shape[-1] * j[1].shape[-1] for j in flat)


def tiles():
    pend = []
    for jobs in blocks:
        ops.gemm_tn_group(jobs, pend)
    ops.gemm_tn_group_reduce(pend)


t_tile = graph_time(tiles)
print(f"M={M} x{NBLK} trunk blocks: tile kernel + batched reduce {t_tile*1e6:7.1f} us ({fl/t_tile/1e12:4.0f} TF)", flush=True)
cnt = ops.tn_blocks_count(flat)
print("blocks:", cnt, "CUs:", ops.num_cus(), flush=True)
if cnt >= 1:
    t_blk = graph_time(lambda: ops.gemm_tn_blocks(flat))
    print(f"M={M} x{NBLK} trunk blocks: whole-block kernel           {t_blk*1e6:7.1f} us ({fl/t_blk/1e12:4.0f} TF)", flush=True)
