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

