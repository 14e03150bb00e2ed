"""ped_gather on the three PED stage shapes of the batch-8 step, timed inside a hipGraph (20 back-to-back launches).
usage: [SPG_LIBRARY=...] python tools/ped_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops


def timeit(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def main():
    dt, dev, B = torch.bfloat16, "cuda", int(os.environ.get("B", "8"))
    gen = torch.Generator(device=dev).manual_seed(0)
    edge = torch.randn(B, 48, 48, 64, device=dev, generator=gen).to(dt)
    for (hc, cin, ec) in ((48, 256, 64), (96, 256, 64), (192, 128, 0)):
        x = torch.randn(B, hc, hc, cin, device=dev, generator=gen).to(dt)
        ss = torch.cat([torch.ones(cin, device=dev), torch.zeros(cin, device=dev)])
        us = timeit(lambda: ops.ped_gather(x, ss, B, hc, hc, cin, edge if ec else None, 48, 48, ec))
        out_mb = B * (2 * hc) ** 2 * (cin + ec) * 2 / 1e6
        in_mb = (x.numel() + (edge.numel() if ec else 0)) * 2 / 1e6
        print(f"{hc:4d} -> {2 * hc:4d}  C {cin}+{ec}: {us:7.1f} us   {in_mb:6.1f} MB in, {out_mb:6.1f} MB out  -> {(in_mb + out_mb) / us * 1e-3 * 1e3:6.2f} GB/ms", flush=True)


if __name__ == "__main__":
    main()
