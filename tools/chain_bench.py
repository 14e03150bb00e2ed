"""Dev tool: spg_nt_chain (dependent GEMMs of a trunk block in one persistent launch) against the same problems as separate gemm_nt
launches: results (bit comparison) and hipGraph timing.  usage: python tools/chain_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spegnet_amd import ops


def timeit(fn, iters=10, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ops.begin_zero_pool("cuda", 1 << 16)
            for _ in range(iters):
                fn()
            ops.end_zero_pool()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
    C = 576
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc)
    x = rn(M, C).to(dt)
    x1 = rn(M, C).to(dt)
    w1, b1 = rn(4 * C, C, sc=C ** -0.5).to(dt), rn(4 * C)
    w2, b2 = rn(C, 4 * C, sc=(4 * C) ** -0.5).to(dt), rn(C)
    # ---- forward MLP: fc1 (+GELU, saves gelu') -> fc2 (+residual)
    hp_a, hp_b = torch.empty(M, 4 * C, dtype=dt, device="cuda"), torch.empty(M, 4 * C, dtype=dt, device="cuda")

    def sep():
        gq = ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_a)
        return gq, ops.gemm_nt(gq, w2, bias=b2, residual=x1)

    def chain():
        return ops.gemm_chain([dict(x=x, w=w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_b),
                               dict(x=None, w=w2, bias=b2, residual=x1)])
    ga, ya = sep()
    gb, yb = chain()
    torch.cuda.synchronize()
    ops.chain_check("cuda")
    for nm, a, b in (("gelu out", ga, gb), ("gelu'", hp_a, hp_b), ("fc2 out", ya, yb)):
        d = (a.float() - b.float()).abs().max().item()
        print(f"fwd {nm:9s}: bit-identical {torch.equal(a, b)}  max |diff| {d:.3e}  (max |ref| {a.float().abs().max().item():.3f})")
    ref = torch.nn.functional.gelu(x.float() @ w1.float().t() + b1).to(dt).float() @ w2.float().t() + b2 + x1.float()
    print(f"fwd chain vs fp32 torch: rel err {((yb.float() - ref).abs().max() / ref.abs().max()).item():.3e}")
    ts, tc = timeit(lambda: sep()), timeit(lambda: chain())
    fl = 2.0 * M * C * 4 * C * 2
    print(f"fwd MLP  M={M}: separate {ts[0]*1e6:7.1f} us (median {ts[1]*1e6:.1f})   chain {tc[0]*1e6:7.1f} us (median {tc[1]*1e6:.1f})   "
          f"{fl/ts[0]/1e12:.0f} -> {fl/tc[0]/1e12:.0f} TFLOP/s")
    if "only_parts" in sys.argv:
        gq = ga
        t1 = timeit(lambda: ops.gemm_chain([dict(x=x, w=w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_b)]))
        t2 = timeit(lambda: ops.gemm_chain([dict(x=gq, w=w2, bias=b2, residual=x1)]))
        s2 = timeit(lambda: ops.gemm_nt(gq, w2, bias=b2, residual=x1))
        print(f"{os.environ.get('SPG_LIBRARY', 'product')}: chain[fc1] {t1[0]*1e6:.1f}  chain[fc2] {t2[0]*1e6:.1f}  gemm_nt fc2 {s2[0]*1e6:.1f}")
        return
    if "parts" in sys.argv:
        gq = ga
        t1 = timeit(lambda: ops.gemm_chain([dict(x=x, w=w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_b)]))
        t2 = timeit(lambda: ops.gemm_chain([dict(x=gq, w=w2, bias=b2, residual=x1)]))
        t3 = timeit(lambda: ops.gemm_chain([dict(x=x, w=w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_b), dict(x=gq, w=w2, bias=b2, residual=x1)]))
        s1 = timeit(lambda: ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, preact_out=hp_a))
        s2 = timeit(lambda: ops.gemm_nt(gq, w2, bias=b2, residual=x1))
        print(f"parts: chain[fc1] {t1[0]*1e6:.1f}  chain[fc2] {t2[0]*1e6:.1f}  chain[fc1, fc2 independent] {t3[0]*1e6:.1f}  |  gemm_nt fc1 {s1[0]*1e6:.1f}  gemm_nt fc2 {s2[0]*1e6:.1f}")
    # ---- backward MLP: dfc2 (x saved gelu') -> dfc1
    dy = rn(M, C).to(dt)
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()      # [4C, C] and [C, 4C]: dgrad weights ([N,K] with N = input features)

    def sepb():
        dh = ops.gemm_nt(dy, w2t, gelu_h=hp_a, act=ops.ACT_MUL_H)
        return dh, ops.gemm_nt(dh, w1t)

    def chainb():
        return ops.gemm_chain([dict(x=dy, w=w2t, gelu_h=hp_a, act=ops.ACT_MUL_H), dict(x=None, w=w1t)])
    da, ea = sepb()
    db, eb = chainb()
    torch.cuda.synchronize()
    ops.chain_check("cuda")
    for nm, a, b in (("dh", da, db), ("dln2", ea, eb)):
        d = (a.float() - b.float()).abs().max().item()
        print(f"bwd {nm:9s}: bit-identical {torch.equal(a, b)}  max |diff| {d:.3e}  (max |ref| {a.float().abs().max().item():.3f})")
    ts, tc = timeit(lambda: sepb()), timeit(lambda: chainb())
    print(f"bwd MLP  M={M}: separate {ts[0]*1e6:7.1f} us (median {ts[1]*1e6:.1f})   chain {tc[0]*1e6:7.1f} us (median {tc[1]*1e6:.1f})   "
          f"{fl/ts[0]/1e12:.0f} -> {fl/tc[0]/1e12:.0f} TFLOP/s")
    # repeated launches must agree bit for bit
    y2 = chain()[1]
    torch.cuda.synchronize()
    print("chain repeat bit-identical:", torch.equal(yb, y2))


if __name__ == "__main__":
    main()
