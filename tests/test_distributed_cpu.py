"""CPU tests (gloo, world_size 2) of the data-parallel gradient path and of the flat-arena host logic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spegnet_amd.engine.arena import ALIGN, Arena, backward_order, group_of
from spegnet_amd.engine.distributed import GradSync, make_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_make_buckets_partition():
    ends = [100, 250, 260, 900, 1000, 1024]
    for be in (1, 150, 500, 5000):
        b = make_buckets(ends, be)
        assert b[0][0] == 0 and b[-1][1] == ends[-1]
        for (s0, e0), (s1, e1) in zip(b, b[1:]):
            assert e0 == s1 and s0 < e0
        assert all(e in ends for _, e in b)          # buckets end on unit boundaries only
        assert all(e - s >= be for s, e in b[:-1])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    n = 4096
    g = torch.randn(n)
    mine = g.clone()
    ends = [512, 1024, 1536, 3000, 4096]
    sync = GradSync(g, ends, bucket_mb=1024 * 4 / (1024 * 1024))  # 1024-element buckets
    # units finish in order; ready() may be called with any monotone offsets, including repeats
    for e in (512, 512, 1536, 1024, 3000):
        sync.ready(e)
    scale = sync.finish()
    allg = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(allg, mine)
    want = sum(allg)
    ok = torch.allclose(g, want, atol=1e-6) and abs(scale - 1.0 / world) < 1e-12
    # a second step re-uses the object
    g.copy_(mine)
    sync.ready(4096)
    sync.finish()
    ok = ok and torch.allclose(g, want, atol=1e-6)
    q.put((rank, bool(ok), len(sync.buckets)))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(nb >= 3 for _, _, nb in res)


def test_arena_layout_and_groups():
    from spegnet_amd.models import SPEGNet
    m = SPEGNet({"encoder": {"variant": "test_tiny"}})
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    ar = Arena(m)
    names = [n for n, _ in m.named_parameters()]
    assert ar.size % ALIGN == 0 and set(ar.offsets) == set(names)
    # values survive, params and grads are views into the arena, every param starts on a 256-element boundary
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n])
        o = ar.offsets[n]
        assert o % ALIGN == 0
        assert p.data_ptr() == ar.p.data_ptr() + 4 * o and p.grad.data_ptr() == ar.g.data_ptr() + 4 * o
    # readiness order: head first, trunk blocks last-to-first, embeddings last
    order = backward_order(names)
    first_enc = min(i for i, n in enumerate(order) if n.startswith("encoder."))
    assert all(not n.startswith("encoder.") for n in order[:first_enc])
    blk = [int(n.split("blocks.")[1].split(".")[0]) for n in order if ".blocks." in n]
    assert blk == sorted(blk, reverse=True)
    assert order[-1].startswith("encoder.encoder.patch_embed") or order[-1].startswith("encoder.encoder.pos_embed")
    assert ar.unit_ends[-1] == ar.size and ar.unit_ends == sorted(ar.unit_ends)
    assert len(ar.unit_ends) == 1 + 8 + 1   # head, 8 blocks of the test trunk, embeddings
    # reference param-group rules (engine/trainer.py:274-306) incl. the Sequential-BN weight-decay quirk
    assert group_of("encoder.encoder.blocks.3.attn.qkv.weight") == 0
    assert group_of("encoder.encoder.blocks.3.norm1.weight") == 1
    assert group_of("context.reduce.1.weight") == 2          # BN inside nn.Sequential: decays
    assert group_of("fusion.bn.weight") == 3 and group_of("decoder.decoder_blocks.0.bn1.bias") == 3
    assert group_of("decoder.decoder_blocks.0.conv1.weight") == 2
    chunk = ar.offsets["fusion.bn.weight"] // ALIGN
    assert int(ar.group_of_chunk[chunk]) == 3
    ar.set_hyper(1e-4, 1e-5, 0.05)
    assert torch.allclose(ar.lr, torch.tensor([5e-6, 5e-6, 1e-4, 1e-4])) and torch.allclose(ar.wd, torch.tensor([0, 0, 1e-5, 0]))


def test_state_dict_keys_match_reference_names():
    from spegnet_amd.models import SPEGNet
    from oracle import spegnet_oracle as O
    m = SPEGNet({"encoder": {"variant": "large", "config_path": "configs/sam2.1/sam2.1_hiera_l.yaml", "checkpoint_path": None}})
    sd, ref = m.state_dict(), O.init_state_dict(0)
    assert list(sorted(sd)) == list(sorted(ref))
    assert all(sd[k].shape == ref[k].shape for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 215_442_100 and m.encoder.param_count == 212_149_296
    assert m.encoder.channels == [144, 288, 576, 1152] and m.in_channels_list == [288, 576, 1152]
    assert m.encoder.get_output_shapes(384, 384) == [(144, 96, 96), (288, 48, 48), (576, 24, 24), (1152, 12, 12)]
    with pytest.raises(ValueError):
        m.encoder.get_output_shapes(100, 384)


def test_product_path_refuses_cpu_and_bad_shapes():
    from spegnet_amd.models import SPEGNet
    m = SPEGNet({"encoder": {"variant": "test_tiny"}})
    with pytest.raises(ValueError, match="divisible by 32"):
        m(torch.zeros(1, 3, 48, 48))
    with pytest.raises(ValueError, match="4D"):
        m(torch.zeros(3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    import inspect, spegnet_amd.models.engine as E, spegnet_amd.models.spegnet as SP, spegnet_amd.ops as OP
    for mod in (E, SP, OP):
        assert "oracle" not in inspect.getsource(mod).replace("the CPU oracle", ""), f"{mod.__name__} must not touch oracle/"


# ---- multi-GPU hygiene of the Trainer's host side (world 2 over gloo; the reference is single-GPU: utils/data_loader.py:287-301,
# ---- engine/trainer.py:559-579 -- sharding, one writer per job and a job-wide validation score are SURVEY 8(e)'s additions) ----
def _make_dataset(root, n=10, size=24):
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(0)
    for sub in ("Imgs", "GT", "Edges"):
        os.makedirs(os.path.join(root, "train", sub), exist_ok=True)
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, "train", "Imgs", f"s{i:02d}.jpg"))
        for sub in ("GT", "Edges"):
            Image.fromarray((rng.random((size, size)) > 0.5).astype(np.uint8) * 255).save(os.path.join(root, "train", sub, f"s{i:02d}.png"))


_MODEL_CFG = {"image_processing": {"target_size": 32, "normalize_mean": [0.485, 0.456, 0.406], "normalize_std": [0.229, 0.224, 0.225]}}


def _shard_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spegnet_amd.engine.trainer import TrainingMonitor, _rank, _world
    from spegnet_amd.utils.data_loader import get_training_loaders
    tr, va = get_training_loaders([root], _MODEL_CFG, batch_size=2, num_workers=0, val_ratio=0.2, rank=rank, world=world, drop_last=True)
    seen = {}
    for ep in (0, 1):
        tr.sampler.set_epoch(ep)
        seen[ep] = [tuple(b["images"].shape) + (float(b["images"].sum()),) for b in tr]
    nval = sum(len(b["masks"]) for b in va)
    # job-wide means: each rank contributes its own shard's sums
    class DM:
        run_dir = os.path.join(root, f"run")
    mon = TrainingMonitor(DM())
    mon.update_batch({"loss": torch.tensor(float(rank + 1))}, 3 + rank)
    mon.all_reduce(torch.device("cpu"))
    mean = mon.end_epoch(0, "val")["loss"]
    q.put((rank, _rank(), _world(), seen, nval, mean, os.path.exists(os.path.join(DM.run_dir, "metrics.json"))))
    dist.barrier()
    dist.destroy_process_group()


def test_training_loader_shards_and_monitor_reduces(tmp_path):
    root = str(tmp_path / "ds")
    _make_dataset(root, n=10)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, rk0, w0, seen0, nv0, m0, f0), (r1, rk1, w1, seen1, nv1, m1, f1) = res
    assert (rk0, rk1, w0, w1) == (0, 1, 2, 2)
    # 8 training samples -> 4 per rank -> 2 full batches of 2 each, every batch full (drop_last), shards disjoint, epochs reshuffled
    for ep in (0, 1):
        assert len(seen0[ep]) == len(seen1[ep]) == 2 and all(s[0] == 2 for s in seen0[ep] + seen1[ep])
        assert not ({s[-1] for s in seen0[ep]} & {s[-1] for s in seen1[ep]}), "ranks must not train on the same samples"
    assert {s[-1] for s in seen0[0]} != {s[-1] for s in seen0[1]}, "set_epoch must change the permutation"
    assert nv0 == nv1 == 1                                       # 2 validation samples, one per rank
    want = (1.0 * 3 + 2.0 * 4) / 7.0
    assert abs(m0 - want) < 1e-12 and abs(m1 - want) < 1e-12     # identical job-wide mean on both ranks
    assert f0 or f1                                              # rank 0 wrote metrics.json ...
    assert os.path.exists(os.path.join(root, "run", "metrics.json"))


def test_monitor_best_key_is_the_early_stop_key():
    from spegnet_amd.engine.trainer import TrainingMonitor
    m = TrainingMonitor(None)
    assert m.check_best_model({"weighted_f": 0.5, "s_alpha": 0.9})
    assert not m.check_best_model({"weighted_f": 0.4, "s_alpha": 0.95})      # a better S_alpha alone is not "best"
    assert m.check_best_model({"weighted_f": 0.6, "s_alpha": 0.1})
    m2 = TrainingMonitor(None)
    assert m2.check_best_model({"loss": 2.0}) and m2.check_best_model({"loss": 1.0}) and not m2.check_best_model({"loss": 1.5})


def test_lib_refuses_a_stale_abi(monkeypatch):
    import __graft_entry__ as g
    g.build()
    from spegnet_amd import _lib
    lib = _lib.load()
    assert lib.spg_version() == _lib.ABI_VERSION
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "spegnet_hip.h")).read()
    assert int(re.search(r"#define\s+SPG_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI revision"):
        _lib.load()


def test_validation_shards_are_unpadded_and_complete():
    """Validation / test shards count every sample exactly once job-wide (DistributedSampler would pad each shard to equal length with
    repeated samples, which the all-reduced validation mean then counts twice)."""
    from spegnet_amd.utils.data_loader import _StridedShard, _loader
    for n, world in ((10, 4), (7, 2), (3, 8), (64, 8)):
        shards = [list(_StridedShard(n, r, world)) for r in range(world)]
        flat = sorted(i for s_ in shards for i in s_)
        assert flat == list(range(n)), (n, world, shards)
        assert max(len(s_) for s_ in shards) - min(len(s_) for s_ in shards) <= 1
    ds = list(range(10))
    ld = _loader(ds, 4, False, 0, rank=1, world=4)
    assert isinstance(ld.sampler, _StridedShard) and list(ld.sampler) == [1, 5, 9]
    tr = _loader(ds, 4, True, 0, rank=1, world=4, drop_last=True)
    from torch.utils.data.distributed import DistributedSampler
    assert isinstance(tr.sampler, DistributedSampler)
