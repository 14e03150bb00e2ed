#!/usr/bin/env python
"""SPEGNet train-step benchmark on MI355X (BASELINE.json metric: img/s fwd+bwd @384x384 bf16).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full optimisation step of the reference's Trainer._process_batch (engine/trainer.py:308-427) on a
synthetic batch of 8 images/GPU at 384x384 (BASELINE.json configs[1]): forward (Hiera-L trunk + CFI + EFE + PED) in
bf16 on the HIP kernels, CODLoss, backward, global-norm clip + AdamW on fp32 master weights, weight re-pack.  Weak
scaling: every rank keeps 8 images; ranks only exchange the gradient all-reduce (RCCL).
Rank 0 prints ONE JSON line (see README / DESIGN.md for the roofline and cpu_baseline legs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PER_GPU_BATCH = 8
S = 384
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def synthetic(B, S, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.randn(B, 3, S, S, generator=g, device=device)
    masks = (torch.rand(B, 1, S, S, generator=g, device=device) > 0.7).float()
    edges = (torch.rand(B, 1, S, S, generator=g, device=device) > 0.95).float()
    return images, masks, edges


def host_cpu():
    """(threads this process may use, CPU model string)"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup CPU quota (the GPU box gives each job a share of the host's cores)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    if os.environ.get("SPG_CPU_THREADS"):
        cores = min(cores, int(os.environ["SPG_CPU_THREADS"]))
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return cores, model


def cpu_baseline():
    """The CPU oracle (fp32 restatement of the reference path, oracle/spegnet_oracle.py) timed on this box's host cores, every core the
    job may use: (1) BASELINE config #2's step -- one full train step at batch 8 @384x384, after one untimed warm-up step; (2) config #1 --
    eval-mode forward of one 384x384 image (what Predictor.predict_single runs, reference engine/predictor.py:336-338), after the
    reference's own warm-up forward (predictor.py:283-288).  A bounded sample (~20-40 s): a reported baseline, not the target."""
    from oracle import spegnet_oracle as O
    cores, model = host_cpu()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} threads of {model}")
    sd = O.init_state_dict(seed=0)
    x1 = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        O.spegnet_forward(sd, x1, training=False)          # warm-up
        t0 = time.time()
        n1 = 3
        for _ in range(n1):
            O.spegnet_forward(sd, x1, training=False)
        dt1 = (time.time() - t0) / n1
    B = PER_GPU_BATCH
    x, masks, edges = O.synthetic_batch(B, S, seed=0)
    st = {}
    O.train_step(sd, st, x, masks, edges)                  # warm-up step (allocator, thread pool)
    t0 = time.time()
    O.train_step(sd, st, x, masks, edges)
    dt = time.time() - t0
    return {"value": round(B / dt, 4), "unit": "img/s", "cores": cores, "cpu_model": model, "kind": "port",
            "sample": f"1 fp32 train step (fwd+CODLoss+bwd+clip+AdamW) of the CPU oracle, batch {B} @{S}x{S}, {dt:.1f} s, after 1 warm-up step",
            "forward_b1": {"value": round(1.0 / dt1, 3), "unit": "img/s", "ms": round(dt1 * 1e3, 1),
                           "sample": f"eval forward of one {S}x{S} image (BASELINE config #1), mean of {n1} after 1 warm-up"}}


PEAK_HBM_GBS = 8000.0       # HBM3E peak, MI355X_MICROARCH.md "Chip-level parameters" (6.29 TB/s measured copy rate)


def roofline_table(rec, bracket_s, dtype):
    """Per-op roofline rows from ops.PROFILE records (name, bound, algorithmic work, start, end): achieved = sum of work / sum of
    HIP-event durations (bracket cost subtracted), against the MFMA peak of the dtype or the HBM peak."""
    peak_mfma = PEAK_BF16_TFLOPS if dtype == "bf16" else 157.3
    tot = {}
    for name, bound, work, e0, e1 in rec:
        a = tot.setdefault((name, bound), [0.0, 0.0, 0])
        a[0] += work; a[1] += max(e0.elapsed_time(e1) * 1e-3 - bracket_s, 1e-7); a[2] += 1
    rows = []
    for (name, bound), (work, sec, n) in tot.items():
        if bound == "mfma":
            ach, peak, unit = work / sec / 1e12, peak_mfma, "TFLOP/s"
        else:
            ach, peak, unit = work / sec / 1e9, PEAK_HBM_GBS, "GB/s"
        rows.append({"kernel": name, "bound": bound, "launches_per_step": n / 2, "avg_us": round(sec / n * 1e6, 2),
                     "ms_per_step": round(sec / 2 * 1e3, 3), "work_per_launch": work / n, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                     "frac": round(ach / peak, 4)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def committed_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    FETCH_SIZE x2 per MI355X_MICROARCH.md; bench.py cannot collect counters itself).  Returns (bytes, source) or (None, None)."""
    path = os.path.join(ROOT, "profiles", "round4_pmc_traffic.json")
    try:
        d = json.load(open(path))
        e = d["kernels"][kernel_key]
        return e["fetch_bytes_x2"] + e["write_bytes"], "profiles/round4_pmc_traffic.json: " + d.get("source", "")
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU (default: BASELINE config)")
    ap.add_argument("--size", type=int, default=S)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches (bucketed all-reduce overlaps backward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--infer", action="store_true",
                    help="BASELINE config 3 instead of the train step: eval-mode forward under one hipGraph (use --batch 64); "
                         "not the headline metric, never the default")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="one GPU, the N > 1 step: a one-rank RCCL process group with every collective of the multi-GPU step issued "
                         "(segmented hipGraphs, per-range all-reduce on the communication stream, bf16 staging casts); measures what the "
                         "structure itself costs before any link is involved.  Not the headline metric, never the default")
    args = ap.parse_args()
    if args.rehearse_comm:
        os.environ["SPG_DIST_FORCE_INIT"] = "1"

    from spegnet_amd.engine.distributed import GradSync, init_process_group_from_env
    rank, world, local = init_process_group_from_env("cuda")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run"
    local = local % max(torch.cuda.device_count(), 1)   # (rehearsal mode: several ranks may share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from spegnet_amd import ops
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.models import SPEGNet
    from spegnet_amd.utils.loss_functions import CODLoss

    if args.infer:
        log(f"rank {rank}/{world} building model (inference)")
        model = SPEGNet({"encoder": {"variant": "large"}, "compute_dtype": args.dtype, "init_seed": 0}).to(dev).eval()
        images = synthetic(args.batch, args.size, dev, seed=1000 * rank)[0]
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(images)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            from spegnet_amd.engine.distributed import graph_capture_mode
            with torch.cuda.graph(graph, capture_error_mode=graph_capture_mode()):     # (N > 1: RCCL's watchdog thread must stay legal during capture)
                out = model(images)
            for _ in range(max(args.warmup, 1)):
                graph.replay()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        assert bool(torch.isfinite(out["predictions"][2].float()).all()), "non-finite logits"
        if rank == 0:
            print(json.dumps({
                "metric": f"img/s inference @{args.size}x{args.size} {args.dtype}", "value": round(world * args.batch * args.steps / dt, 3),
                "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"SPEGNet eval-mode forward (Hiera-L trunk + CFI + EFE + PED), batch {args.batch}/GPU "
                                       f"@{args.size}x{args.size}, random-init weights, one hipGraph replay per step",
                           "global_batch": world * args.batch, "image_size": args.size, "parallelism": f"dp{world}", "launch": "hipGraph"}}))
        return

    log(f"rank {rank}/{world} building model")
    model = SPEGNet({"encoder": {"variant": "large"}, "compute_dtype": args.dtype, "init_seed": 0}).to(dev).train()
    arena = Arena(model)
    model.mark_params_changed()
    arena.set_hyper(1e-4, 1e-5, 0.05)                      # configs/default.yaml:25-28 of the reference
    crit = CODLoss(scale_weights=[0.2, 0.3, 0.5], boundary_weight=2.0, bce_weight=1.25, iou_weight=1.0, edge_weight=0.75,
                   edge_focal_alpha=0.75, edge_focal_gamma=2.0).to(dev)  # configs/default.yaml:34-43
    sync = GradSync(arena.g, arena.unit_ends, force=args.rehearse_comm) if (world > 1 or args.rehearse_comm) else None
    step = TrainStep(model, crit, arena, grad_clip=1.0, sync=sync, capture=not args.no_graph)
    images, masks, edges = synthetic(args.batch, args.size, dev, seed=1000 * rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 1)):
        try:
            losses = step(images, masks, edges)
        except Exception as e:  # graph capture problem (e.g. a collective that cannot be combined with capture): run eager
            if args.no_graph or i > 0:
                raise
            log(f"graph mode failed ({type(e).__name__}: {e}); falling back to eager launches with overlapped all-reduce")
            torch.cuda.synchronize()
            args.no_graph = True
            step = TrainStep(model, crit, arena, grad_clip=1.0, sync=sync, capture=False)
            losses = step(images, masks, edges)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done (includes hipGraph capture)" if not args.no_graph else "first eager step done")
    barrier()
    log("warm-up done, timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step(images, masks, edges)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    log(f"timed {args.steps} steps in {dt:.3f} s")
    loss_val = float(losses["loss"])
    assert loss_val == loss_val, "loss is NaN"

    # ---- roofline leg: the dominant kernel (gemm_nt, bf16 dense MFMA GEMM) timed with HIP events on its own stream over
    #      instrumented eager steps run right after the timed region (hipGraph replays cannot carry per-kernel events).
    roof = None
    if rank == 0 and not args.no_roofline and not args.rehearse_comm:
        eager = TrainStep(model, crit, arena, grad_clip=1.0, sync=None, capture=False) if world == 1 else None
        if eager is not None:
            # Eager launches are host-bound: without help the GPU idles between kernels and each start event fires long before
            # its kernel arrives, so the measured interval includes the host gap.  A calibrated spin kernel ahead of each
            # instrumented step keeps the GPU busy while the host queues the whole step; the events then bracket back-to-back
            # execution, which is what rocprofv3's per-kernel durations report too.
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record(); torch.cuda._sleep(20_000_000); ev1.record()
            torch.cuda.synchronize()
            spin_cycles_per_ms = 20_000_000 / max(ev0.elapsed_time(ev1), 1e-3)
            # cost of the bracket itself (two event records with nothing between them), measured under the same condition and
            # subtracted from every interval
            torch.cuda._sleep(int(20 * spin_cycles_per_ms))
            pairs = []
            for _ in range(64):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); b.record(); pairs.append((a, b))
            torch.cuda.synchronize()
            gaps = sorted(a.elapsed_time(b) * 1e-3 for a, b in pairs)
            empty_s = gaps[len(gaps) // 2]
            # ... but around a KERNEL part of that cost overlaps the kernel's own execution (the empty pair over-corrected the
            # round-1/2 lines by ~2 us against rocprofv3).  Two-point calibration instead: intervals around spin kernels of n and 2n
            # cycles are E1 = b + T and E2 = b + 2T, so b = 2 E1 - E2 without knowing T.
            n_spin = int(0.02 * spin_cycles_per_ms)        # ~20 us
            torch.cuda._sleep(int(20 * spin_cycles_per_ms))
            p1, p2 = [], []
            for _ in range(48):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); torch.cuda._sleep(n_spin); b.record(); p1.append((a, b))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); torch.cuda._sleep(2 * n_spin); b.record(); p2.append((a, b))
            torch.cuda.synchronize()
            e1 = sorted(a.elapsed_time(b) * 1e-3 for a, b in p1)[len(p1) // 2]
            e2 = sorted(a.elapsed_time(b) * 1e-3 for a, b in p2)[len(p2) // 2]
            bracket_s = min(max(2 * e1 - e2, 0.0), empty_s)
            ops.PROFILE = []
            for _ in range(2):
                torch.cuda._sleep(int(150 * spin_cycles_per_ms))
                eager(images, masks, edges)
                torch.cuda.synchronize()
            rec, ops.PROFILE = ops.PROFILE, None
            rows = roofline_table(rec, bracket_s, args.dtype)
            want = "gemm_nt<%s,dense>" % ("bf16" if args.dtype == "bf16" else "f32")
            dom = next(r for r in rows if r["kernel"] == want)
            traffic, src = committed_traffic(want)
            roof = {"bound": "mfma", "kernel": "gemm_nt<%s,dense>: gemm_nt_pipe_kernel + gemm_nt_v3_kernel (all dense gemm_nt launches: Linear / 1x1 conv fwd + dgrad)" % args.dtype,
                    "achieved": dom["achieved"], "peak": dom["peak"], "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic,
                    "traffic_source": src, "launches": int(dom["launches_per_step"] * 2), "avg_launch_us": dom["avg_us"],
                    "event_bracket_us": round(bracket_s * 1e6, 2), "event_empty_pair_us": round(empty_s * 1e6, 2), "flop_per_launch_avg": dom["work_per_launch"],
                    # every instrumented op of the step (2 eager steps, HIP events on the launch stream): recompute frac = achieved / peak,
                    # achieved = work_per_launch / avg_us
                    "kernels": rows}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse_comm:
        cpu = cpu_baseline()

    if rank == 0:
        n_img = world * args.batch * args.steps
        line = {
            "metric": "img/s fwd+bwd @384x384 bf16" if (args.size == 384 and args.dtype == "bf16") else f"img/s fwd+bwd @{args.size}x{args.size} {args.dtype}",
            "value": round(n_img / dt, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SPEGNet (Hiera-L trunk + CFI + EFE + PED) train step: fwd + CODLoss + bwd + clip + AdamW, "
                                   f"batch {args.batch}/GPU @{args.size}x{args.size}, random-init weights",
                       "global_batch": world * args.batch, "image_size": args.size, "parallelism": f"dp{world}",
                       # (step.capture turns False when the hipGraph capture failed on some rank and every rank agreed to run eagerly)
                       "launch": "eager, bucketed all-reduce overlapped with backward" if (args.no_graph or not step.capture) else ("hipGraph" if (world == 1 and not args.rehearse_comm) else f"hipGraph segments (fwd+loss+bwd in {getattr(step, 'n_segments', 8)} pieces) with RCCL all-reduce (bf16 payload) of finished gradient ranges on a side stream | hipGraph optimizer"),
                       "final_loss": round(loss_val, 5), **({"rehearsal": "one-rank RCCL group, every collective of the N > 1 step issued"} if args.rehearse_comm else {})},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1 or args.rehearse_comm:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
