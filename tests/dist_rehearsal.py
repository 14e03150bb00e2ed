"""Two-rank rehearsal of the multi-GPU train step on ONE GPU (gloo backend; RCCL needs one GPU per rank).  Launched by
tests/test_distributed_gpu.py as two child processes BEFORE the parent touches the GPU.

Each rank trains the tiny SPEGNet (fp32 compute) -- or, with `large`, the full Hiera-L block table in bf16 compute at 64 px -- on ITS OWN batch with the segmented hipGraph step (forward + loss + backward captured in segments, the
all-reduce of finished gradient ranges between them on a side stream, bf16 payload or fp32) and, for comparison, rank 0 recomputes what
data parallelism must produce: the gradients of both batches from two single-rank eager passes, averaged, one clip + AdamW step.
Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "graph"          # graph | eager
    payload = sys.argv[2] if len(sys.argv) > 2 else "fp32"         # fp32 | bf16
    variant = sys.argv[3] if len(sys.argv) > 3 else "tiny"         # tiny (fp32 compute, 128 px) | large (bf16 compute, Hiera-L block table, 64 px)
    from oracle import spegnet_oracle as O
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.distributed import GradSync, init_process_group_from_env
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.models import SPEGNet
    from spegnet_amd.utils.loss_functions import CODLoss
    import torch.distributed as dist
    rank, world, local = init_process_group_from_env("cuda")
    torch.cuda.set_device(0)
    large = variant == "large"
    cfg = O.HIERA_L if large else O.HIERA_TINY_TEST
    sd = O.init_state_dict(seed=3, cfg=cfg)

    def fresh():
        m = SPEGNet({"encoder": {"variant": "large" if large else "test_tiny"}, "compute_dtype": "bf16" if large else "fp32", "init": "empty"})
        m.load_state_dict(sd)
        m = m.cuda().train()
        ar = Arena(m)
        m.mark_params_changed()
        ar.set_hyper(1e-3, 1e-2, 0.5)
        return m, ar

    batches = [O.synthetic_batch(2, 64, seed=80 + r) if large else O.synthetic_batch(4, 128, seed=80 + r) for r in range(world)]
    dev = lambda b: (b[0].cuda(), torch.stack(b[1]).cuda(), torch.stack(b[2]).cuda())
    m, ar = fresh()
    # (one rank with a forced process group: the RCCL rehearsal of test_distributed_gpu.py -- every collective of the N > 1 step runs)
    sync = GradSync(ar.g, ar.unit_ends, compress_bf16=(payload == "bf16"), force=(world == 1))
    step = TrainStep(m, CODLoss().cuda(), ar, grad_clip=1.0, sync=sync, capture=(mode == "graph"))
    out = step(*dev(batches[rank]))
    torch.cuda.synchronize()
    res = {"rank": rank, "mode": mode, "payload": payload, "variant": variant, "loss": float(out["loss"]), "gnorm": float(ar.gnorm_sq.sqrt()),
           "segments": len(step.segments) if step.segments else 0, "backend": dist.get_backend(), "collectives": bool(step.comm)}
    # every rank must hold the same parameters after the step
    p = ar.p.detach().clone()
    allp = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(allp, p)
    res["ranks_agree"] = bool(all(torch.equal(allp[0], q) for q in allp))
    if rank == 0:
        # reference: single-rank gradients of each rank's batch, averaged, then ONE optimizer step
        gs = []
        for r in range(world):
            mr, arr = fresh()
            crit = CODLoss().cuda()
            o = mr(dev(batches[r])[0])
            ls = crit.forward_batched(o["predictions"], o["edge"], dev(batches[r])[1], dev(batches[r])[2])
            ls["loss"].backward()
            torch.cuda.synchronize()
            gs.append(arr.g.detach().clone())
        mref, aref = fresh()
        aref.g.copy_(sum(gs) / world)
        aref._clean = False
        aref.step(1.0)
        torch.cuda.synchronize()
        p0 = fresh()[1].p.detach().clone()                 # the initial parameters in arena order
        upd, upd_ref = p - p0, aref.p - p0
        thr = 0.25 * float(upd_ref.abs().max())
        res["frac_updates_differ"] = float(((upd - upd_ref).abs() > thr).float().mean())
        res["gnorm_ref"] = float(aref.gnorm_sq.sqrt())
    print("REHEARSAL " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
