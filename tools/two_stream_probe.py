"""Dev probe: does running the trunk as TWO half-batch chains on two HIP streams (parallel branches of one captured graph) beat one
full-batch chain?  Runs the forward of N stage-3 trunk blocks (Hiera-L, 24 x 24 tokens, C = 576) both ways and times graph replays.
usage: python tools/two_stream_probe.py [blocks=12] [B=8] [bwd]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spegnet_oracle as O   # (weights only: a dev tool, not the product path)
from spegnet_amd import ops
from spegnet_amd.models import SPEGNet


def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    bwd = "bwd" in sys.argv
    m = SPEGNet({"encoder": {"variant": "large"}, "compute_dtype": "bf16"}).cuda()
    m.train(True)
    eng = m.engine
    blocks = [b for b in eng.blocks if b["dim"] == 576 and b["dim_out"] == 576 and b["window"] == 16][:nblk]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 24, 24, 576, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(B, 24, 24, 576, device="cuda", generator=g).to(torch.bfloat16)
    eng.block_wgrads = False

    def run(xx, dd, Bq):
        ctxs = []
        for b in blocks:
            xx, c = eng.block_fwd(b, xx, Bq, 24, 24, True)
            ctxs.append(c)
        if bwd:
            d = dd
            for b, c in zip(reversed(blocks), reversed(ctxs)):
                d = eng.block_bwd(b, c, d, Bq)
            eng.flush_ln_params()
            return d
        return xx

    def one():
        return run(x, dy, B)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h = B // 2

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = run(x[:h], dy[:h], h)
        with torch.cuda.stream(s2):
            b_ = run(x[h:], dy[h:], B - h)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return a, b_

    def timeit(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                ops.begin_zero_pool("cuda", 8 << 20)
                fn()
                ops.end_zero_pool()
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); gr.replay(); e1.record(st)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
        return min(ts), sorted(ts)[len(ts) // 2]
    t1 = timeit(one)
    t2 = timeit(two)
    t1b = timeit(one)
    print(f"{nblk} stage-3 blocks {'fwd+bwd' if bwd else 'fwd'}, batch {B}: one stream {t1[0]*1e3:.0f} us (median {t1[1]*1e3:.0f}; again {t1b[0]*1e3:.0f})   "
          f"two half-batch streams {t2[0]*1e3:.0f} us (median {t2[1]*1e3:.0f})   per block {t1[0]*1e3/nblk:.1f} -> {t2[0]*1e3/nblk:.1f} us")


if __name__ == "__main__":
    main()
