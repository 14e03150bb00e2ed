"""Would warming the NEXT GEMM's weights (L2 / Infinity Cache) during the kernel before it pay?  36 stage-3-like trunk blocks with their own
weights (287 MB in bf16: more than the 256 MB Infinity Cache, so every weight read is cold as in the step), LN -> qkv -> proj -> LN -> fc1 ->
fc2 per block, captured as one hipGraph and replayed.  Mode `touch`: a read of the next GEMM's weight matrix (torch.sum) is launched before
the kernel that precedes that GEMM.  Run under rocprofv3 --kernel-trace --stats and compare the GEMM kernels' average durations (the touch
launches themselves are extra kernels here: only the GEMM durations matter).
usage: rocprofv3 --kernel-trace --stats -d out -o run -- python3 tools/prefetch_probe.py [touch | hint]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops


def main():
    touch = len(sys.argv) > 1 and sys.argv[1] == "touch"
    hint = len(sys.argv) > 1 and sys.argv[1] == "hint"          # the product mechanism: every GEMM requests the next GEMM's weights on its way in
    dt, dev = torch.bfloat16, "cuda"
    M, C, NB = 4608, 576, 36
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda n, k: (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(dt)
    hintln = len(sys.argv) > 1 and sys.argv[1] == "hintln"      # ... and the LayerNorm parameters the kernel after it reads (cold fp32, 4.6 KB)
    biaswarm = len(sys.argv) > 1 and sys.argv[1] == "biaswarm"  # hint + the four (cold, fp32) bias vectors read one kernel ahead
    hint = hint or hintln or biaswarm

    def with_ln(n, k):
        """weight [n, k] bf16 followed in the SAME allocation by an fp32 gamma / beta pair: one hint range covers both"""
        buf = torch.empty(n * k * 2 + 2 * C * 4, dtype=torch.uint8, device=dev)
        w = buf[:n * k * 2].view(dt).view(n, k)
        w.copy_(mk(n, k))
        gb = buf[n * k * 2:].view(torch.float32)
        gb[:C] = 1.0; gb[C:] = 0.0
        return buf, w, gb[:C], gb[C:]
    blocks = []
    for _ in range(NB):
        qb, qw, g1, b1 = with_ln(3 * C, C)          # norm1's parameters ride behind the qkv weight
        fb, fw, g2, b2 = with_ln(4 * C, C)          # norm2's behind fc1
        blocks.append(dict(qkv=qw, proj=mk(C, C), fc1=fw, fc2=mk(C, 4 * C), g1=g1, b1=b1, g2=g2, b2=b2, qkv_buf=qb, fc1_buf=fb,
                           bq=torch.zeros(3 * C, device=dev), bp=torch.zeros(C, device=dev), b1f=torch.zeros(4 * C, device=dev), b2f=torch.zeros(C, device=dev),
                           pad=torch.zeros(1 << 20, device=dev)))      # (4 MB between the blocks' biases: no two share a page)
    x0 = torch.randn(M, C, device=dev, generator=g).to(dt)
    sink = torch.zeros(1, device=dev)

    def warm(w):
        if touch:
            sink.add_(w.float().sum() * 0)       # (reads the whole matrix; two small torch kernels)

    def warmb(b):
        if biaswarm:
            sink.add_(b.sum() * 0)

    def chain():
        x = x0
        for i, b in enumerate(blocks):
            kq, kf = ("qkv_buf", "fc1_buf") if hintln else ("qkv", "fc1")
            nxt = blocks[i + 1][kq] if i + 1 < len(blocks) else None
            H = (lambda t: t) if hint else (lambda t: None)
            warm(b["qkv"])
            ln, _, _ = ops.layernorm_fwd(x, b["g1"], b["b1"], 1e-6)
            warmb(b["bq"]); warmb(b["bp"]); warmb(b["b1f"]); warmb(b["b2f"])
            ln, _, _ = ops.layernorm_fwd(x, b["g1"], b["b1"], 1e-6)      # (again: the bias reads above sit two kernels ahead of their first use)
            qkv = ops.gemm_nt(ln, b["qkv"], bias=b["bq"], prefetch=H(b["proj"]))
            warm(b["proj"])
            a = qkv[:, :C].contiguous()                                  # (stands in for the attention kernel: a kernel between the two GEMMs)
            x1 = ops.gemm_nt(a, b["proj"], bias=b["bp"], residual=x, prefetch=H(b[kf]))
            warm(b["fc1"])
            ln2, _, _ = ops.layernorm_fwd(x1, b["g2"], b["b2"], 1e-6)
            h = ops.gemm_nt(ln2, b["fc1"], bias=b["b1f"], act=ops.ACT_GELU, prefetch=H(b["fc2"]))
            warm(b["fc2"])
            hh = h * 1                                                    # (a kernel between fc1 and fc2 so the warm-up has something to hide behind)
            x = ops.gemm_nt(hh, b["fc2"], bias=b["b2f"], residual=x1, prefetch=H(nxt) if nxt is not None else None)
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
    print(f"mode {sys.argv[1] if len(sys.argv) > 1 else 'plain'}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per chain of {NB} blocks", flush=True)


if __name__ == "__main__":
    main()
