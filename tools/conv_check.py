"""Correctness + timing of the bf16 3x3 convolution path of spg_gemm_nt (conv_halo.hip vs the implicit GEMM) on the PED / EFE shapes.
usage: [SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so SPG_CONV_HALO=0|1|64|128|256] python tools/conv_check.py [time] [big]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spegnet_amd import ops

# (B, H, W, Ci, Co, tag)
SMALL = [(1, 8, 32, 64, 64, "one tile"), (2, 9, 33, 64, 64, "ragged"), (1, 16, 64, 128, 128, "2x2 tiles kc2"), (2, 24, 40, 64, 256, "bn256"),
         (1, 40, 72, 320, 64, "kc5"), (3, 17, 31, 128, 320, "co320"), (2, 48, 48, 256, 64, "efe"), (1, 200, 96, 64, 128, "many tiles")]
BIG = [(8, 48, 48, 256, 64, "EFE fwd"), (8, 48, 48, 64, 256, "EFE dgrad"),
       (8, 96, 96, 320, 256, "s1 conv1"), (8, 96, 96, 256, 256, "s1 conv2 / dgrad"), (8, 96, 96, 256, 320, "s1 conv1 dgrad"),
       (8, 192, 192, 320, 128, "s2 conv1"), (8, 192, 192, 128, 128, "s2 conv2 / dgrad"), (8, 192, 192, 128, 320, "s2 conv1 dgrad"),
       (8, 384, 384, 128, 64, "s3 conv1"), (8, 384, 384, 64, 64, "s3 conv2 / dgrad"), (8, 384, 384, 64, 128, "s3 conv1 dgrad")]


def timeit(fn, iters=10):
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
        for _ in range(3):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


def main():
    dt = torch.bfloat16
    do_time = "time" in sys.argv
    shapes = SMALL + (BIG if ("big" in sys.argv or do_time) else [])
    only = os.environ.get("CONV_ONLY")
    if only:
        shapes = [s_ for s_ in BIG if any(o in s_[5] for o in only.split(","))]
    nocheck = os.environ.get("CONV_NOCHECK") == "1"
    g = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    tot_t, tot_f = 0.0, 0.0
    for B, H, W_, Ci, Co, tag in shapes:
        x = torch.randn(B, H, W_, Ci, device="cuda", generator=g).to(dt)
        w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (9 * Ci) ** -0.5
        wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(dt)
        bias = torch.randn(Co, device="cuda", generator=g)
        out = torch.full((B * H * W_, Co), float("nan"), device="cuda", dtype=dt)
        ops.gemm_nt(x, wp, bias=bias, conv=(B, H, W_, Ci), out=out)
        torch.cuda.synchronize()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wp.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2), bias, padding=1)
        ref = ref.permute(0, 2, 3, 1).reshape(-1, Co)
        bad = ~torch.isfinite(out.float())
        e = float((out.float() - ref).abs().max() / ref.abs().max()) if not bad.any() else float("inf")
        if nocheck:
            e = 0.0
        worst = max(worst, e)
        line = f"{tag:18s} B{B} {H}x{W_} {Ci}->{Co} err {e:.2e}" + (f" NONFINITE {int(bad.sum())}" if bad.any() else "")
        if e > 2.5e-2:
            d = (out.float() - ref).abs().view(B, H, W_, Co)
            idx = (d > 2.5e-2 * ref.abs().max()).nonzero()
            line += f" | {len(idx)} bad, first {idx[:4].tolist()}"
        if do_time and (B, H, W_, Ci, Co, tag) in BIG:
            t = timeit(lambda: ops.gemm_nt(x, wp, bias=bias, conv=(B, H, W_, Ci), out=out))
            fl = 2.0 * B * H * W_ * Co * 9 * Ci
            tot_t += t; tot_f += fl
            line += f" | {t*1e6:8.1f}us {fl/t/1e12:6.0f} TF"
        print(line, flush=True)
    if tot_t:
        print(f"sum of timed shapes: {tot_t*1e3:.3f} ms, {tot_f/tot_t/1e12:.0f} TF")
    if "wgrad" in sys.argv:
        from spegnet_amd import _lib
        WG = [(8, 48, 48, 256, 64, "EFE"), (8, 96, 96, 320, 256, "s1 conv1"), (8, 96, 96, 256, 256, "s1 conv2"), (8, 192, 192, 320, 128, "s2 conv1"),
              (8, 192, 192, 128, 128, "s2 conv2"), (8, 384, 384, 128, 64, "s3 conv1"), (8, 384, 384, 64, 64, "s3 conv2")]
        tn, to, tf = 0.0, 0.0, 0.0
        for B, H, W_, Ci, Co, tag in WG:
            x = torch.randn(B, H, W_, Ci, device="cuda", generator=g).to(dt)
            dy = torch.randn(B, H, W_, Co, device="cuda", generator=g).to(dt)
            M, K = B * H * W_, 9 * Ci
            dw_new, dw_old = torch.zeros(Co, K, device="cuda"), torch.zeros(Co, K, device="cuda")
            db = torch.zeros(Co, device="cuda")
            ops.gemm_tn(dy, x, dw_new, conv=(B, H, W_, Ci), dbias=db)
            wsb = _lib.load().spg_gemm_tn_workspace_bytes(1, M, Co, K)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
            old = lambda: _lib.call("spg_gemm_tn", 1, dy.data_ptr(), x.data_ptr(), dw_old.data_ptr(), None, ws.data_ptr(), wsb, M, Co, K, Co, Ci, K,
                                    1, B, H, W_, Ci, 0, torch.cuda.current_stream().cuda_stream)
            old()
            torch.cuda.synchronize()
            e = float((dw_new - dw_old).abs().max() / dw_old.abs().max())
            t_new = timeit(lambda: ops.gemm_tn(dy, x, dw_new, conv=(B, H, W_, Ci), dbias=db))
            t_old = timeit(old)
            fl = 2.0 * M * Co * K
            tn += t_new; to += t_old; tf += fl
            print(f"wgrad {tag:9s} B{B} {H}x{W_} {Ci}->{Co} new vs old rel diff {e:.1e} | halo {t_new*1e6:8.1f}us {fl/t_new/1e12:5.0f} TF | implicit GEMM {t_old*1e6:8.1f}us {fl/t_old/1e12:5.0f} TF", flush=True)
        print(f"wgrad sum: halo {tn*1e3:.3f} ms ({tf/tn/1e12:.0f} TF), implicit GEMM {to*1e3:.3f} ms ({tf/to/1e12:.0f} TF)")
    print("worst", worst)
    assert worst < 2.5e-2, worst


if __name__ == "__main__":
    main()
