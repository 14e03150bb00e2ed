"""Would warming the NEXT GEMM's weights (L2 / Infinity Cache) during the kernel before it pay?  36 stage-3-like trunk blocks with their own
weights (287 MB in bf16: more than the 256 MB Infinity Cache, so every weight read is cold as in the step), LN -> qkv -> proj -> LN -> fc1 ->
fc2 per block, captured as one hipGraph and replayed.  Mode `touch`: a read of the next GEMM's weight matrix (torch.sum) is launched before
the kernel that precedes that GEMM.  Run under rocprofv3 --kernel-trace --stats and compare the GEMM kernels' average durations (the touch
launches themselves are extra kernels here: only the GEMM durations matter).
usage: rocprofv3 --kernel-trace --stats -d out -o run -- python3 tools/prefetch_probe.py [touch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops


def main():
    touch = len(sys.argv) > 1 and sys.argv[1] == "touch"
    dt, dev = torch.bfloat16, "cuda"
    M, C, NB = 4608, 576, 36
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda n, k: (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(dt)
    blocks = [dict(qkv=mk(3 * C, C), proj=mk(C, C), fc1=mk(4 * C, C), fc2=mk(C, 4 * C), g1=torch.ones(C, device=dev), b1=torch.zeros(C, device=dev))
              for _ in range(NB)]
    x0 = torch.randn(M, C, device=dev, generator=g).to(dt)
    sink = torch.zeros(1, device=dev)

    def warm(w):
        if touch:
            sink.add_(w.float().sum() * 0)       # (reads the whole matrix; two small torch kernels)

    def chain():
        x = x0
        for b in blocks:
            warm(b["qkv"])
            ln, _, _ = ops.layernorm_fwd(x, b["g1"], b["b1"], 1e-6)
            qkv = ops.gemm_nt(ln, b["qkv"])
            warm(b["proj"])
            a = qkv[:, :C].contiguous()                                  # (stands in for the attention kernel: a kernel between the two GEMMs)
            x1 = ops.gemm_nt(a, b["proj"], residual=x)
            warm(b["fc1"])
            ln2, _, _ = ops.layernorm_fwd(x1, b["g1"], b["b1"], 1e-6)
            h = ops.gemm_nt(ln2, b["fc1"], act=ops.ACT_GELU)
            warm(b["fc2"])
            hh = h * 1                                                    # (a kernel between fc1 and fc2 so the warm-up has something to hide behind)
            x = ops.gemm_nt(hh, b["fc2"], residual=x1)
        return x

    for _ in range(2):
        chain()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            chain()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            gr.replay()
        e1.record(st)
        torch.cuda.synchronize()
    print(f"mode {'touch' if touch else 'plain'}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per chain of {NB} blocks", flush=True)


if __name__ == "__main__":
    main()
