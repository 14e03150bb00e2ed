"""GPU rehearsal of the N > 1 train step: two ranks on ONE GPU over gloo (the only way to execute the multi-rank code path on a 1-GPU
box; RCCL needs a GPU per rank).  This file sorts first among the `-m gpu` files on purpose: the ranks are started as child processes
BEFORE this process has initialised the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, payload, world=2, variant="tiny", backend="gloo"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SPG_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", SPG_DIST_FORCE_INIT="1" if world == 1 else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_rehearsal.py"), mode, payload, variant], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    res = {}
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        line = [ln for ln in o.splitlines() if ln.startswith("REHEARSAL ")]
        assert line, o[-2000:]
        res[r] = json.loads(line[-1][len("REHEARSAL "):])
    return res


@pytest.mark.parametrize("mode,payload", [("graph", "fp32"), ("graph", "bf16"), ("eager", "fp32")])
def test_two_rank_step_equals_averaged_single_rank_gradients(mode, payload):
    """Segmented-hipGraph (and eager bucketed) data-parallel step on two ranks with DIFFERENT batches: every rank ends with the same
    parameters, and they are the parameters one AdamW step on the average of the two single-rank gradients gives."""
    res = _run(mode, payload)
    assert all(v["ranks_agree"] for v in res.values()), res
    r0 = res[0]
    if mode == "graph":
        assert r0["segments"] >= 4, r0
    tol = 0.02 if payload == "fp32" else 0.08        # Adam's first step is ~lr*sign(g): a few near-zero gradients may flip
    assert r0["frac_updates_differ"] < tol, res
    # (arena.gnorm_sq is the norm of the SUMMED gradient; the 1/world scale is applied inside the optimizer kernel)
    assert abs(r0["gnorm"] / 2 - r0["gnorm_ref"]) < (2e-3 if payload == "fp32" else 2e-2) * r0["gnorm_ref"], res


def test_two_rank_step_bf16_compute_hiera_large_block_table():
    """The same rehearsal on the shipping configuration's code path: bf16 compute, the 48-block Hiera-L table (stage widths 144..1152,
    global-attention blocks, all four pooling transitions), 64 px inputs, bf16 wire payload -- the segment boundaries and bucket ends are
    the ones bench.py --gpus N uses."""
    res = _run("graph", "bf16", variant="large")
    assert all(v["ranks_agree"] for v in res.values()), res
    r0 = res[0]
    assert r0["segments"] >= 4, r0
    assert r0["loss"] == r0["loss"]
    assert r0["frac_updates_differ"] < 0.10, res
    assert abs(r0["gnorm"] / 2 - r0["gnorm_ref"]) < 3e-2 * r0["gnorm_ref"], res


@pytest.mark.parametrize("mode,payload,variant", [("graph", "bf16", "large"), ("graph", "fp32", "tiny"), ("eager", "bf16", "tiny")])
def test_one_rank_rccl_executes_the_multi_gpu_choreography(mode, payload, variant):
    """RCCL itself (backend "nccl"), on the one GPU a test box has: a ONE-rank process group with GradSync(force=True), so the step
    issues every collective of the N > 1 path -- the per-segment all-reduces on the communication stream between hipGraph replays, the
    bf16 staging casts, the capture-outcome agreement -- through RCCL.  The sum over one rank is the identity: the parameters must
    equal a plain single-GPU step's (within the bf16 payload's rounding)."""
    res = _run(mode, payload, world=1, variant=variant, backend="nccl")
    r0 = res[0]
    assert r0["backend"] == "nccl" and r0["collectives"], r0
    if mode == "graph":
        assert r0["segments"] >= 4, r0
    assert r0["loss"] == r0["loss"]
    assert r0["frac_updates_differ"] < (0.02 if payload == "fp32" else 0.10), r0
    assert abs(r0["gnorm"] - r0["gnorm_ref"]) < (2e-3 if payload == "fp32" else 3e-2) * r0["gnorm_ref"], r0
