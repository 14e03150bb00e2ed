"""Trainer / Predictor entry points on the GPU (reference surface: engine/trainer.py, engine/predictor.py)."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import spegnet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "spegnet_amd", "configs", "default.yaml")))
    cfg["model"]["encoder"].update(variant="test_tiny", checkpoint_path=None)
    cfg["model"]["image_processing"]["target_size"] = 64
    cfg["training"].update(batch_size=3, capture_graph=False, use_amp=False)
    return cfg


def test_predictor_matches_oracle(tmp_path):
    from PIL import Image
    from spegnet_amd.engine.predictor import Predictor
    cfg = _cfg()
    sd = O.init_state_dict(seed=2, cfg=O.HIERA_TINY_TEST)
    ck = tmp_path / "model_best.pth"
    mc = dict(cfg["model"], compute_dtype="fp32")
    torch.save({"model_state_dict": sd, "config": {"model": mc}}, ck)
    rng = np.random.RandomState(0)
    img = tmp_path / "img.png"
    Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(img)
    pr = Predictor(str(ck), mc, dir_manager=None, device="cuda", batch_size=1)
    seg, edge, orig = pr.predict_single(str(img), output_size=(90, 120))
    assert seg.shape == (90, 120) and edge.shape == (90, 120) and orig.shape == (90, 120, 3)
    x = pr.image_processor(str(img)).image[None]
    with torch.no_grad():
        ref = O.spegnet_forward(sd, x, training=False, cfg=O.HIERA_TINY_TEST)
        rs = torch.nn.functional.interpolate(ref["predictions"][-1], size=(90, 120), mode="bilinear", align_corners=False).sigmoid()[0, 0]
    assert float((torch.from_numpy(seg) - rs).abs().max()) < 1e-3
    with pytest.raises(FileNotFoundError):
        Predictor(str(tmp_path / "nope.pth"), mc, None, device="cuda")
    out = pr.predict_directory(str(tmp_path))
    assert out["total_predictions"] == 2


def test_trainer_process_batch_ragged_and_fixed():
    from spegnet_amd.engine.trainer import Trainer
    cfg = _cfg()
    tr = Trainer(cfg, dir_manager=None, device=torch.device("cuda"))
    x, masks, edges = O.synthetic_batch(3, 64, seed=7)
    m1, t1 = tr._process_batch({"images": x, "masks": masks, "edges": edges}, is_train=True)
    assert set(m1) == {"loss", "seg_loss", "edge_loss"} and "batch_time" in t1
    # ragged ground truth takes the per-sample resize path of the reference (trainer.py:358-383)
    masks2 = [torch.rand(1, 50 + 7 * i, 70) .gt(0.7).float() for i in range(3)]
    edges2 = [torch.rand(1, 50 + 7 * i, 70).gt(0.95).float() for i in range(3)]
    m2, _ = tr._process_batch({"images": x, "masks": masks2, "edges": edges2}, is_train=True)
    assert float(m2["loss"]) == float(m2["loss"])
    m3, _ = tr._process_batch({"images": x, "masks": masks, "edges": edges}, is_train=False)
    assert float(m3["loss"]) > 0
    groups = tr._get_param_groups()
    assert [g["weight_decay"] for g in groups] == [0.0, 0.0, 1e-5, 0.0] and abs(groups[0]["lr"] - 5e-6) < 1e-12


def test_image_processor_device_path_matches_host_path(tmp_path):
    """CODImageProcessor.process_image_device (decode on the host, one HIP kernel for /255 + antialiased resize + normalise) returns
    what process_image (the reference's torch arithmetic on the CPU) returns for the same file."""
    import numpy as np
    from PIL import Image
    from spegnet_amd.utils.image_processor import CODImageProcessor
    g = torch.Generator().manual_seed(5)
    px = torch.randint(0, 256, (333, 517, 3), generator=g, dtype=torch.uint8)
    path = tmp_path / "img.png"
    Image.fromarray(px.numpy()).save(str(path))
    proc = CODImageProcessor(target_size=384)
    host = proc.process_image(path)
    dev = proc.process_image_device(path)
    assert dev.is_cuda and dev.shape == host.shape == (3, 384, 384)
    assert float((dev.cpu() - host).abs().max()) < 5e-6



def _write_dataset(root, n, S=(72, 96), seed=0):
    """root/<name>/train/{Imgs,GT,Edges} + test/{Imgs,GT} with n synthetic samples each"""
    from PIL import Image
    rng = np.random.RandomState(seed)
    d = root / "TOY"
    for split, with_edges in (("train", True), ("test", False)):
        for sub in ("Imgs", "GT") + (("Edges",) if with_edges else ()):
            (d / split / sub).mkdir(parents=True, exist_ok=True)
        for i in range(n):
            h, w = S[0] + 8 * (i % 3), S[1] + 4 * (i % 2)
            Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(d / split / "Imgs" / f"s{i}.jpg")
            Image.fromarray((rng.rand(h, w) > 0.6).astype(np.uint8) * 255).save(d / split / "GT" / f"s{i}.png")
            if with_edges:
                Image.fromarray((rng.rand(h, w) > 0.9).astype(np.uint8) * 255).save(d / split / "Edges" / f"s{i}.png")
    return str(d)


def test_predict_batch_is_batched_and_equals_predict_single(tmp_path):
    """Predictor.predict_batch: a chunk of batch_size images is ONE forward (hipGraph for full chunks, eager for the short last one) fed
    by the batched device preprocessing; every image's maps equal predict_single's."""
    from PIL import Image
    from spegnet_amd.engine.predictor import Predictor
    cfg = _cfg()
    sd = O.init_state_dict(seed=2, cfg=O.HIERA_TINY_TEST)
    ck = tmp_path / "model_best.pth"
    mc = dict(cfg["model"], compute_dtype="bf16")
    torch.save({"model_state_dict": sd, "config": {"model": mc}}, ck)
    rng = np.random.RandomState(1)
    paths = []
    for i in range(7):                                       # 7 images, batch 3: two graph replays + one short eager batch
        p = tmp_path / f"im{i}.png"
        Image.fromarray(rng.randint(0, 255, (80 + 5 * i, 100 + 3 * i, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))

    class Keep:
        def __init__(self): self.items, self.t = {}, {}
        def update_timing(self, k, v): self.t.setdefault(k, []).append(v)
        def log_message(self, m): pass
        def save_prediction(self, name, seg, edge, orig): self.items[name] = (seg.copy(), edge.copy(), orig.shape)
        def summarize(self): return {"total_predictions": len(self.items), "forwards": len(self.t.get("inference", []))}

    keep = Keep()
    pr = Predictor(str(ck), mc, dir_manager=None, device="cuda", batch_size=3, result_manager=keep)
    out = pr.predict_batch(paths, output_size=(64, 64))
    assert out["total_predictions"] == 7 and out["forwards"] == 3, out
    assert pr._graph is not None
    for p in paths:
        seg, edge, orig = pr.predict_single(p, output_size=(64, 64))
        bs, be, shp = keep.items[os.path.basename(p)]
        assert shp == orig.shape
        assert float(np.abs(bs - seg).max()) < 2e-3 and float(np.abs(be - edge).max()) < 2e-3, p


def test_predict_batch_at_config3_batch_size(tmp_path):
    """BASELINE config #3's shape of work through the Predictor (reference engine/predictor.py:376-430): batch_size 64 -> one captured
    forward for the full chunk of 64 images, an eager one for the remaining 6; spot-checked against predict_single."""
    from PIL import Image
    from spegnet_amd.engine.predictor import Predictor
    cfg = _cfg()
    sd = O.init_state_dict(seed=2, cfg=O.HIERA_TINY_TEST)
    ck = tmp_path / "model_best.pth"
    mc = dict(cfg["model"], compute_dtype="bf16")
    torch.save({"model_state_dict": sd, "config": {"model": mc}}, ck)
    rng = np.random.RandomState(3)
    paths = []
    for i in range(70):
        p = tmp_path / f"im{i:02d}.png"
        Image.fromarray(rng.randint(0, 255, (40 + i % 7, 48 + i % 5, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))

    class Keep:
        def __init__(self): self.items, self.t = {}, {}
        def update_timing(self, k, v): self.t.setdefault(k, []).append(v)
        def log_message(self, m): pass
        def save_prediction(self, name, seg, edge, orig): self.items[name] = (seg.copy(), edge.copy(), orig.shape)
        def summarize(self): return {"total_predictions": len(self.items), "forwards": len(self.t.get("inference", []))}

    keep = Keep()
    pr = Predictor(str(ck), mc, dir_manager=None, device="cuda", batch_size=64, result_manager=keep)
    out = pr.predict_batch(paths, output_size=(64, 64))
    assert out["total_predictions"] == 70 and out["forwards"] == 2, out
    single = Predictor(str(ck), mc, dir_manager=None, device="cuda", batch_size=1)
    for i in (0, 37, 63, 69):
        seg, edge, _ = single.predict_single(paths[i], output_size=(64, 64))
        name = [k for k in keep.items if k.startswith(f"im{i:02d}")][0]
        assert float(np.abs(keep.items[name][0] - seg).max()) < 2e-2 and float(np.abs(keep.items[name][1] - edge).max()) < 2e-2


def test_trainer_train_loop_with_loader_device_preprocess_and_resume(tmp_path):
    """Trainer.train(dataset_dirs) end to end on a synthetic on-disk dataset: the package's own loader (reference surface), original-size
    (ragged, non-square) ground truth -> the per-sample loss path, device preprocessing + prefetch, checkpoints, and resume()."""
    from spegnet_amd.engine.trainer import Trainer

    class Dirs:
        def __init__(self, d): self.run_dir = d; self.checkpoint_dir = d
    # (8 samples, 25% validation -> 6 training samples in batches of 2: like the reference's loader there is no drop_last, and a last
    # batch of ONE fails train-mode BatchNorm in the global e-ASPP branch exactly as torch does)
    root = _write_dataset(tmp_path, 8)
    cfg = _cfg()
    cfg["training"].update(batch_size=2, num_epochs=2, num_workers=0, val_ratio=0.25, save_freq=1, device_preprocess=True, early_stop_patience=5)
    tr = Trainer(cfg, dir_manager=Dirs(str(tmp_path / "run")), device=torch.device("cuda"))
    p0 = tr.arena.p.detach().clone()
    tr.train([root])
    assert not torch.equal(p0, tr.arena.p) and bool(torch.isfinite(tr.arena.p).all())
    ck = tmp_path / "run" / "checkpoint_001.pth"
    assert ck.exists()
    # resume: a new trainer restored from the checkpoint continues from identical optimizer state
    tr2 = Trainer(cfg, dir_manager=Dirs(str(tmp_path / "run2")), device=torch.device("cuda"))
    nxt = tr2.resume(str(ck))
    assert nxt == 2
    assert torch.equal(tr2.arena.m, tr.arena.m) and torch.equal(tr2.arena.v, tr.arena.v) and torch.equal(tr2.arena.step_f, tr.arena.step_f)
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), tr2.model.state_dict().items()):
        assert torch.equal(a, b), k
    x, masks, edges = O.synthetic_batch(2, 64, seed=3)
    a = tr._process_batch({"images": x, "masks": masks, "edges": edges}, is_train=True)[0]
    b = tr2._process_batch({"images": x, "masks": masks, "edges": edges}, is_train=True)[0]
    assert float(a["loss"]) == float(b["loss"])
    # one more step from the restored state reproduces the original run (bias gradients summed by the wgrad GEMMs' float atomics may
    # differ in their last bits: compare to 1e-6 of the update scale)
    assert float((tr.arena.p - tr2.arena.p).abs().max()) < 1e-6 * float(tr.arena.p.abs().max())


def test_fused_step_gating_and_captured_shape_change():
    """Equal-size but NON-square / non-multiple ground truth must take the per-sample path (the fused HIP loss needs square targets that are
    a multiple of every prediction size), and a captured step must not swallow a batch of another shape."""
    from spegnet_amd.engine.trainer import Trainer
    cfg = _cfg()
    cfg["training"].update(batch_size=3, capture_graph=True)
    tr = Trainer(cfg, dir_manager=None, device=torch.device("cuda"))
    x, masks, edges = O.synthetic_batch(3, 64, seed=7)
    m1, _ = tr._process_batch({"images": x, "masks": masks, "edges": edges}, is_train=True)          # captured
    assert tr.step_fn.graph is not None
    rect = [torch.rand(1, 48, 80).gt(0.7).float() for _ in range(3)]
    m2, _ = tr._process_batch({"images": x, "masks": rect, "edges": [r.clone() for r in rect]}, is_train=True)   # same size, not square
    odd = [torch.rand(1, 100, 100).gt(0.7).float() for _ in range(3)]                               # square, but 100 % 64 != 0
    m3, _ = tr._process_batch({"images": x, "masks": odd, "edges": [r.clone() for r in odd]}, is_train=True)
    short, _ = tr._process_batch({"images": x[:2], "masks": masks[:2], "edges": edges[:2]}, is_train=True)   # short last batch: eager
    for m in (m1, m2, m3, short):
        assert float(m["loss"]) == float(m["loss"]) and float(m["loss"]) > 0


@pytest.mark.parametrize("H,W", [(64, 96), (96, 64)])
def test_non_square_input_matches_oracle(H, W):
    """reference models/feature_encoding.py:230-233 accepts any H, W % 32 == 0"""
    from spegnet_amd.models import SPEGNet
    cfg = O.HIERA_TINY_TEST
    sd = O.init_state_dict(seed=3, cfg=cfg)
    m = SPEGNet({"encoder": {"variant": "test_tiny"}, "compute_dtype": "fp32"})
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = O.spegnet_forward(sd, x, training=False, cfg=cfg)
        out = m(x.cuda())
    for a, b in zip(out["predictions"] + [out["edge"]], ref["predictions"] + [ref["edge"]]):
        assert a.shape == b.shape
        assert float((a.float().cpu() - b).abs().max() / b.abs().max()) < 1e-3


def test_evaluator_on_device_metrics(tmp_path):
    """Evaluator.evaluate (reference engine/evaluator.py:395-468 surface): batched forward, per-sample resize to the original ground-truth
    size, the five COD measures computed on the device, category bucketing and evaluation_summary.json."""
    import json
    from spegnet_amd.engine.evaluator import Evaluator
    from spegnet_amd.utils.data_loader import get_test_loaders
    from spegnet_amd.utils import metrics as M

    class Dirs:
        def __init__(self, d): self.run_dir = d
    root = _write_dataset(tmp_path, 5)
    cfg = _cfg()
    sd = O.init_state_dict(seed=2, cfg=O.HIERA_TINY_TEST)
    ck = tmp_path / "model_best.pth"
    mc = dict(cfg["model"], compute_dtype="fp32")
    torch.save({"model_state_dict": sd, "config": {"model": mc}}, ck)
    loaders = get_test_loaders([root], mc, batch_size=2, num_workers=0)
    assert list(loaders) == ["TOY"]
    ev = Evaluator(str(ck), Dirs(str(tmp_path / "eval")), mc, torch.device("cuda"), batch_size=2)
    res = ev.evaluate(loaders["TOY"], "TOY")
    assert set(res) == {"s_alpha", "weighted_f", "mae", "e_phi", "mean_f"} and all(0.0 <= v <= 1.0 for v in res.values())
    assert sum(ev.categories["TOY"].values()) == 5
    summ = json.load(open(tmp_path / "eval" / "TOY" / "evaluation_summary.json"))
    assert summ["timing"]["total_samples"] == 5 and abs(summ["metrics"]["mae"] - res["mae"]) < 1e-12
    # first sample recomputed by hand: oracle forward -> resize -> sigmoid(sigmoid(.)) (the reference's evaluation quirk) -> metrics
    b = next(iter(loaders["TOY"]))
    with torch.no_grad():
        ref = O.spegnet_forward(sd, b["images"][:1], training=False, cfg=O.HIERA_TINY_TEST)["predictions"][-1]
        z = torch.nn.functional.interpolate(ref, size=b["masks"][0].shape[-2:], mode="bilinear", align_corners=False).sigmoid()
        want = M.MetricsProcessor().compute_metrics(z, [b["masks"][0]])
    ev2 = Evaluator(str(ck), None, mc, torch.device("cuda"), batch_size=1)
    got, n = ev2._process_batch("TOY", {"images": b["images"][:1], "masks": b["masks"][:1], "names": b["names"][:1]})
    assert n == 1
    for k in want:
        assert abs(got[k] - want[k]) < 2e-3, (k, got[k], want[k])
