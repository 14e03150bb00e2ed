"""How much of a backward kernel's time is its COLD saved activations?  36 stage-3 blocks' worth of saved tensors (so none survives in the
256 MB Infinity Cache between uses), each consumer -- layernorm_bwd (reads the saved x), the windowed attention backward (saved qkv, output,
lse) -- run once per block inside one hipGraph; mode `touch` reads the block's saved tensors one kernel ahead (torch.sum).  Compare the
consumers' average durations under rocprofv3 --kernel-trace --stats.
usage: rocprofv3 --kernel-trace --stats -d out -o run -- python3 tools/cold_probe.py [touch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops


def main():
    touch = len(sys.argv) > 1 and sys.argv[1] == "touch"
    dt, dev = torch.bfloat16, "cuda"
    B, H, W, C, heads, hd, ws, NB = 8, 24, 24, 576, 8, 72, 16, 36
    M = B * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    blocks = []
    for _ in range(NB):
        qkv = rn(B, H, W, 3 * C)
        bias = torch.zeros(3 * C, device=dev, dtype=dt)
        att, lse = ops.attn_fwd(qkv, bias, B, H, W, heads, hd, ws)
        blocks.append(dict(x=rn(M, C), mean=torch.zeros(M, device=dev), rstd=torch.ones(M, device=dev), qkv=qkv, bias=bias, att=att, lse=lse,
                           filler=rn(M, 4 * C)))
    gamma = torch.ones(C, device=dev)
    dbias = torch.zeros(3 * C, device=dev)
    dy0 = rn(M, C)
    sink = torch.zeros(1, device=dev)

    def warm(*ts):
        if touch:
            for t in ts:
                sink.add_(t.float().sum() * 0)

    def chain():
        dy = dy0
        for b in blocks:
            warm(b["x"])
            f = b["filler"] * 1                         # a streaming kernel in between (stands in for the GEMM before the consumer: 42 MB of traffic)
            dx = ops.layernorm_bwd(dy, b["x"], gamma, b["mean"], b["rstd"], None, None)
            warm(b["qkv"], b["att"], b["lse"])
            f2 = b["filler"] * 1
            dqkv, _ = ops.attn_bwd(b["qkv"], b["bias"], b["att"], dx.view(B, H, W, C), b["lse"], dbias, B, H, W, heads, hd, ws)
            dy = dqkv.view(M, 3 * C)[:, :C].contiguous()
        return dy

    for _ in range(2):
        chain()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            chain()
        for _ in range(12):
            gr.replay()
        torch.cuda.synchronize()
    print("done", "touch" if touch else "plain", flush=True)


if __name__ == "__main__":
    main()
