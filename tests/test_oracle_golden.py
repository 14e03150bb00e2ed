"""Pins the CPU oracle against golden vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import sys, os
import pytest
import torch

from conftest import rel_err
from oracle import spegnet_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402  (only its seeded input builders are used; it never imports the reference here)


def _head_sd():
    return {k: v for k, v in O.init_state_dict(seed=11).items() if not k.startswith("encoder.")}


@pytest.mark.parametrize("tag", ["eval_small", "train_small", "eval_384"])
def test_head_matches_reference(golden, tag):
    rec = golden("head.pt")[tag]
    sd = _head_sd()
    assert abs(MG.sd_checksum(sd) - rec["sd_checksum"]) < 1e-6 * rec["sd_checksum"], "seeded init drifted"
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not O.is_buffer_key(k)}
    feats = [f.requires_grad_(True) for f in MG.head_inputs(rec["B"], rec["h"], seed=rec["in_seed"])]
    r = O.head_forward(sd, feats, training=rec["training"])
    if tag == "eval_384":
        assert rel_err(r["predictions"][0], rec["pred1"]) < 1e-5
        assert rel_err(r["predictions"][1][..., ::2, ::2], rec["pred2_s2"]) < 1e-5
        assert rel_err(r["predictions"][2][..., ::4, ::4], rec["pred3_s4"]) < 1e-5
        assert rel_err(r["edge"], rec["edge"]) < 1e-5
        for k in ("fused", "context", "edge_features"):
            assert rel_err(r["features"][k].mean((2, 3)), rec[k + "_chanmean"]) < 1e-5
        return
    for a, b in zip(r["predictions"], rec["predictions"]):
        assert rel_err(a, b) < 1e-5
    assert rel_err(r["edge"], rec["edge"]) < 1e-5
    for k in ("fused", "context", "edge_features"):
        assert rel_err(r["features"][k], rec[k]) < 1e-5
    gw = torch.Generator().manual_seed(rec["w_seed"])
    ws = [torch.randn(p.shape, generator=gw) for p in r["predictions"]] + [torch.randn(r["edge"].shape, generator=gw)]
    sum((w * p).sum() for w, p in zip(ws, r["predictions"] + [r["edge"]])).backward()
    # train-mode BN over B=3 samples (global branch sees 3 values/channel) amplifies fp32 rounding
    gtol = 2e-3 if rec["training"] else 1e-4
    for f, g in zip(feats, rec["grad_inputs"]):
        assert rel_err(f.grad, g) < gtol
    for k, n in rec["grad_norms"].items():
        got = float(params[k].grad.norm()) if params[k].grad is not None else 0.0
        assert abs(got - n) <= gtol * n + (5e-4 if rec["training"] else 1e-5), (k, got, n)  # conv-bias-before-BN grads are ~0 in train mode
    for k, g in rec["grads"].items():
        assert rel_err(params[k].grad, g) < gtol or float(g.abs().max()) < 1e-4, k
    if rec["training"]:
        for k, v in rec["running"].items():
            assert rel_err(sd[k], v) < 1e-5, k
        assert int(sd["fusion.bn.num_batches_tracked"]) == 1


def test_easpp_grouped_channel_quirk():
    """group g of the grouped 1x1 reads concat channels 5g..5g+4 (branch-major)."""
    sd = _head_sd()
    x = torch.randn(2, 512, 6, 6, generator=torch.Generator().manual_seed(1))
    base = O.cfi_easpp(sd, x, False)
    w = sd["context.fusion.0.weight"]
    w2 = w.clone(); w2[25, 3] += 1.0  # group 25, tap 3 -> concat channel 128 = branch 1 channel 0
    sd2 = dict(sd); sd2["context.fusion.0.weight"] = w2
    y = F_reduce = None
    import torch.nn.functional as F
    xr = F.relu(F.batch_norm(F.conv2d(x, sd["context.reduce.0.weight"]), sd["context.reduce.1.running_mean"], sd["context.reduce.1.running_var"], sd["context.reduce.1.weight"], sd["context.reduce.1.bias"], False, 0.1, 1e-5))
    b1 = F.conv2d(xr, sd["context.branches.1.0.weight"], padding=6, dilation=6, groups=128)
    b1 = F.relu(F.batch_norm(b1, sd["context.branches.1.1.running_mean"], sd["context.branches.1.1.running_var"], sd["context.branches.1.1.weight"], sd["context.branches.1.1.bias"], False, 0.1, 1e-5))
    assert (b1[:, 0].abs().sum() > 0)
    diff = (O.cfi_easpp(sd2, x, False) - base).abs().sum()
    assert diff > 0


@pytest.mark.parametrize("cfg", ["yaml", "ctor_default"])
@pytest.mark.parametrize("name", ["rand", "zeros", "ones", "ragged"])
def test_codloss_matches_reference(golden, cfg, name):
    rec = golden("loss.pt")[f"{cfg}/{name}"]
    preds, edge, masks, edges = MG.loss_case(name)
    preds = [p.requires_grad_(True) for p in preds]
    edge = edge.requires_grad_(True)
    kw = dict(O.LOSS_DEFAULT_YAML) if cfg == "yaml" else {}
    ld = O.cod_loss(preds, edge, masks, edges, **kw)
    for k, v in rec["loss"].items():
        assert abs(float(ld[k]) - v) < 2e-6 * max(1.0, abs(v)), (k, float(ld[k]), v)
    ld["loss"].backward()
    for p, g in zip(preds, rec["grad_preds"]):
        assert rel_err(p.grad, g) < 1e-5
    assert rel_err(edge.grad, rec["grad_edge"]) < 1e-5
    bw = kw.get("boundary_weight", 5.0)
    wm = O.boundary_weights(masks[0], bw)
    assert abs(float(wm.double().sum()) - rec["wmap0_sum"]) < 1e-6 * rec["wmap0_sum"]
    assert torch.allclose(wm[:, :6, :6], rec["wmap0_corner"], atol=1e-6)


@pytest.mark.parametrize("tag", ["tiny_64", "tiny_96", "large_64", "large_128", "large_384"])
def test_trunk_matches_hf_crosscheck(golden, tag):
    """Secondary pin (the reference's own trunk lives in the absent `sam2` package)."""
    rec = golden("trunk_hf.pt")[tag]
    cfg = O.HIERA_TINY_TEST if "tiny" in tag else O.HIERA_L
    sd = O.init_state_dict(seed=rec["sd_seed"], cfg=cfg)
    x = torch.randn(rec["B"], 3, rec["S"], rec["S"], generator=torch.Generator().manual_seed(rec["in_seed"]))
    with torch.no_grad():
        feats = O.hiera_trunk(sd, x, cfg=cfg)
    assert rel_err(feats[3], rec["feat_s4"]) < 2e-5
    assert rel_err(feats[2][:, ::8], rec["feat_s3_slice"]) < 2e-5
    assert rel_err(feats[1][:, ::16], rec["feat_s2_slice"]) < 2e-5
    if rec["feats"] is not None:
        for a, b in zip(feats, rec["feats"]):
            assert rel_err(a, b) < 2e-5
    for f, (mu, sdv, mx) in zip(feats, rec["feat_stats"]):
        assert abs(float(f.mean()) - mu) < 1e-4 and abs(float(f.abs().max()) - mx) < 1e-3 * mx


def test_block_table_matches_survey():
    t = O.hiera_block_table(O.HIERA_L)
    assert len(t) == 48
    assert (t[0]["dim"], t[0]["window"]) == (144, 8)
    assert (t[2]["dim"], t[2]["dim_out"], t[2]["window"], t[2]["q_stride"]) == (144, 288, 8, 2)
    assert (t[3]["window"], t[8]["window"], t[8]["dim_out"], t[9]["window"]) == (4, 4, 576, 16)
    assert [t[i]["window"] for i in (23, 33, 43)] == [0, 0, 0]
    assert (t[44]["dim_out"], t[44]["window"], t[45]["window"], t[44]["heads"]) == (1152, 16, 8, 16)
    assert [b["idx"] for b in t if b["stage_end"]] == [1, 7, 43, 47]
    n = sum(v.numel() for k, v in O.init_state_dict(0).items() if k.startswith("encoder.") )
    assert n == 212_149_296


def test_input_validation():
    sd = O.init_state_dict(0, cfg=O.HIERA_TINY_TEST)
    with pytest.raises(ValueError):
        O.hiera_trunk(sd, torch.zeros(1, 3, 48, 64), cfg=O.HIERA_TINY_TEST)
    with pytest.raises(ValueError):
        O.hiera_trunk(sd, torch.zeros(3, 64, 64), cfg=O.HIERA_TINY_TEST)


def test_preprocess_matches_reference_image_processor(golden):
    """oracle.preprocess_image vs what the reference's CODImageProcessor.process_image returned for the same decoded pixels
    (fixture generated by importing utils/image_processor.py: tests/golden/make_golden.py::gen_preprocess)."""
    fx = golden("preprocess.pt")
    for name, c in fx.items():
        out = O.preprocess_image(c["pixels"], c["size"])
        assert out.shape == c["out"].shape
        assert float((out - c["out"]).abs().max()) < 1e-6, name

