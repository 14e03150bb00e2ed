"""Where does the N > 1 step's extra millisecond go?  Times, on one GPU, (a) the single-graph step, (b) the SEGMENTED step without any
communication (force_segmented, no GradSync), (c) the rehearsed step with an fp32 wire (collectives issued, no staging casts), (d) the
rehearsed step as shipped (bf16 wire).  usage: python tools/seg_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SPG_DIST_FORCE_INIT", "1")
import torch
from bench import synthetic
from spegnet_amd.engine.arena import Arena
from spegnet_amd.engine.distributed import GradSync, init_process_group_from_env
from spegnet_amd.engine.trainer import TrainStep
from spegnet_amd.models import SPEGNet
from spegnet_amd.utils.loss_functions import CODLoss


def run(tag, **kw):
    dev = torch.device("cuda", 0)
    model = SPEGNet({"encoder": {"variant": "large"}, "compute_dtype": "bf16", "init_seed": 0}).to(dev).train()
    arena = Arena(model)
    model.mark_params_changed()
    arena.set_hyper(1e-4, 1e-5, 0.05)
    crit = CODLoss().to(dev)
    sync = GradSync(arena.g, arena.unit_ends, force=True, compress_bf16=kw.pop("bf16", True)) if kw.pop("comm", False) else None
    step = TrainStep(model, crit, arena, grad_clip=1.0, sync=sync, capture=True, **kw)
    batch = synthetic(8, 384, dev, seed=0)
    for _ in range(5):
        step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(*batch)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step", flush=True)
    del step, model, arena
    torch.cuda.empty_cache()


init_process_group_from_env("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
tag = f"[segments {os.environ.get('SPG_SEGMENTS', '8')}, comm CUs {os.environ.get('SPG_COMM_CUS', '240')}] "
if which == "all":
    run("single graph")
run(tag + "segmented, no communication", force_segmented=True)
if which == "all":
    run("rehearsed, fp32 wire (no casts)", comm=True, bf16=False)
    run("rehearsed, bf16 wire", comm=True)
    run("single graph")
