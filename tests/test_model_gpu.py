"""End-to-end GPU parity: the HIP SPEGNet path (through the C ABI) against the CPU oracle on identical
seeded weights and inputs.  fp32 mode is the parity mode (tolerance 1e-3 relative, mask bit-exact away from
|logit| < 1e-4); bf16 mode is checked against the same oracle at bf16 tolerances."""
import pytest
import torch

from conftest import rel_err
from oracle import spegnet_oracle as O

pytestmark = pytest.mark.gpu


def make_model(variant, dtype, seed=3, train=False):
    from spegnet_amd.models import SPEGNet
    cfg = O.HIERA_L if variant == "large" else O.HIERA_TINY_TEST
    sd = O.init_state_dict(seed=seed, cfg=cfg)
    m = SPEGNet({"encoder": {"variant": variant if variant == "large" else "test_tiny"}, "compute_dtype": dtype})
    m.load_state_dict(sd)
    m = m.cuda()
    m.train(train)
    return m, sd, cfg


def cmp_outputs(out, ref, tol, tag):
    errs = {}
    for i in range(3):
        errs[f"pred{i + 1}"] = rel_err(out["predictions"][i].float(), ref["predictions"][i])
    errs["edge"] = rel_err(out["edge"].float(), ref["edge"])
    for k in ("context", "fused", "edge_features"):
        errs[k] = rel_err(out["features"][k].float(), ref["features"][k])
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, f"{tag}: {bad} (all: {errs})"
    return errs


@pytest.mark.parametrize("variant,S,B", [("tiny", 64, 2), ("tiny", 96, 1), ("large", 64, 1), ("large", 128, 2)])
def test_forward_eval_fp32_matches_oracle(variant, S, B):
    m, sd, cfg = make_model(variant, "fp32")
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = O.spegnet_forward(sd, x, training=False, cfg=cfg)
        out = m(x.cuda())
        feats = m.encoder(x.cuda())
        rf = O.hiera_trunk(sd, x, cfg=cfg)
    for a, b in zip(feats, rf):
        assert a.shape == b.shape
        assert rel_err(a.float(), b) < 1e-3
    cmp_outputs(out, ref, 1e-3, f"{variant}@{S}")
    # thresholded mask: bit-exact wherever the oracle's logit is not within rounding of zero
    p, r = out["predictions"][2].float().cpu(), ref["predictions"][2]
    sure = r.abs() > 1e-3 * r.abs().max()
    assert torch.equal((p > 0)[sure], (r > 0)[sure])
    assert sure.float().mean() > 0.99


def test_forward_384_fp32_matches_oracle_and_golden(golden):
    m, sd, cfg = make_model("large", "fp32", seed=3)
    x = torch.randn(1, 3, 384, 384, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = O.spegnet_forward(sd, x, training=False, cfg=cfg)
        out = m(x.cuda())
    errs = cmp_outputs(out, ref, 1e-3, "large@384")
    p, r = out["predictions"][2].float().cpu(), ref["predictions"][2]
    sure = r.abs() > 1e-3 * r.abs().max()
    assert torch.equal((p > 0)[sure], (r > 0)[sure])
    assert out["predictions"][2].shape == (1, 1, 384, 384) and out["edge"].shape == (1, 1, 48, 48)
    print("rel errs @384 fp32:", errs)


@pytest.mark.parametrize("variant,S,B", [("tiny", 64, 2), ("large", 128, 1)])
def test_forward_bf16_close_to_oracle(variant, S, B):
    m, sd, cfg = make_model(variant, "bf16")
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = O.spegnet_forward(sd, x, training=False, cfg=cfg)
        out = m(x.cuda())
    cmp_outputs(out, ref, 6e-2, f"bf16 {variant}@{S}")


def oracle_loss_and_grads(sd, cfg, x, masks, edges, loss_cfg, dtype=torch.float32):
    import oracle.spegnet_oracle as OM
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    OM._LAPLACE = OM._LAPLACE.to(dtype)
    x, masks, edges = x.to(dtype), [m.to(dtype) for m in masks], [e.to(dtype) for e in edges]
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not O.is_buffer_key(k)}
    out = O.spegnet_forward(sd, x, training=True, cfg=cfg)
    losses = O.cod_loss(out["predictions"], out["edge"], masks, edges, **loss_cfg)
    grads = torch.autograd.grad(losses["loss"], list(params.values()), allow_unused=True)
    OM._LAPLACE = OM._LAPLACE.float()
    return out, losses, dict(zip(params.keys(), grads)), sd


def _detach_out(o):
    return {"predictions": [t.detach() for t in o["predictions"]], "edge": o["edge"].detach(),
            "features": {a: b.detach() for a, b in o["features"].items()}}


@pytest.mark.parametrize("variant,S,B", [("tiny", 128, 4), ("large", 128, 4)])
def test_train_forward_backward_fp32_matches_oracle(variant, S, B):
    """Train-mode BN makes gradients ill-conditioned in fp32 (few samples per channel), so the yardstick is an fp64
    run of the oracle: the HIP fp32 path must be as close to it as the fp32 oracle itself is (x3 + 2e-3 floor)."""
    from spegnet_amd.utils.loss_functions import CODLoss
    m, sd, cfg = make_model(variant, "fp32", train=True)
    x, masks, edges = O.synthetic_batch(B, S, seed=20)
    ref_out, ref_losses, g32, ref_sd = oracle_loss_and_grads(sd, cfg, x, masks, edges, O.LOSS_DEFAULT_YAML)
    _, l64, g64, _ = oracle_loss_and_grads(sd, cfg, x, masks, edges, O.LOSS_DEFAULT_YAML, torch.float64)
    crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
    out = m(x.cuda())
    cmp_outputs(_detach_out(out), _detach_out(ref_out), 1e-3, "train fwd")
    losses = crit.forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
    for k in ("loss", "seg_loss", "edge_loss"):
        assert abs(float(losses[k]) - float(l64[k])) < 1e-4 * abs(float(l64[k])), k
    # running statistics after exactly one train-mode forward (checked here: the gradient retries below run more forwards)
    st = m.state_dict()
    for k in ("fusion.bn.running_mean", "context.global_branch.2.running_var", "decoder.decoder_blocks.2.bn2.running_var"):
        assert rel_err(st[k].float(), ref_sd[k]) < 1e-3, k
    assert int(st["fusion.bn.num_batches_tracked"]) == 1
    gmax = max(float(g.abs().max()) for g in g64.values() if g is not None)
    e_o32 = {}
    for k, p in m.named_parameters():
        if g64[k] is not None:
            e_o32[k] = float((g32[k].double() - g64[k]).abs().max()) / max(float(g64[k].abs().max()), 1e-3 * gmax)

    def pct(d, q):
        v = sorted(d.values())
        return v[min(len(v) - 1, int(q * len(v)))]
    o80, omed = pct(e_o32, 0.8), pct(e_o32, 0.5)

    # Train-mode BatchNorm on a handful of samples (the e-ASPP global branch normalises B values per channel) is ill-conditioned in
    # fp32: which way the float atomics of a reduction happen to round shifts a normalised activation visibly, and every gradient
    # upstream of it with it.  Identical inputs therefore give a few discrete outcomes (tools/diag_grad_dist.py: median per-parameter
    # error 8e-4, 1.2e-3 or 7e-3 against fp64 -- the fp32 ORACLE's own worst parameter is off by 1.3e-2); the kernels themselves are
    # bit-reproducible (tools/diag_tn_repro.py).  So: every run must stay inside loose bounds, and at least one of three runs must be as
    # close to fp64 as the fp32 oracle is (x3) -- a wrong kernel fails all of them.
    tight_ok, report = False, []
    for attempt in range(3):
        if attempt:
            out = m(x.cuda())
            losses = crit.forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
        for p in m.parameters():
            p.grad = None
        losses["loss"].backward()
        e_hip = {}
        for k, p in m.named_parameters():
            if g64[k] is not None:
                e_hip[k] = float((p.grad.cpu().double() - g64[k]).abs().max()) / max(float(g64[k].abs().max()), 1e-3 * gmax)
        h95, h80, hmed, hmax = pct(e_hip, 0.95), pct(e_hip, 0.8), pct(e_hip, 0.5), max(e_hip.values())
        top = dict(sorted(e_hip.items(), key=lambda kv: -kv[1])[:5])
        report.append(f"run {attempt}: median {hmed:.2e} p80 {h80:.2e} p95 {h95:.2e} max {hmax:.2e}")
        assert h80 < 3e-2 and h95 < 5e-2 and hmax < 0.5, f"gradient error outside the loose bounds: {report[-1]}; top {top}"
        if h80 < 3 * o80 + 2e-3 and hmed < 3 * omed + 5e-4:
            tight_ok = True
            break
    assert tight_ok, f"no run as close to fp64 as the fp32 oracle (x3; oracle median {omed:.2e} p80 {o80:.2e}): {report}"


@pytest.mark.parametrize("capture", [False, True])
def test_train_steps_match_oracle(capture):
    """Two optimizer steps of the fp32 path == two oracle steps (clip + AdamW with the reference's groups)."""
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.utils.loss_functions import CODLoss
    m, sd, cfg = make_model("tiny", "fp32", train=True)
    arena = Arena(m)
    m.mark_params_changed()
    arena.set_hyper(1e-3, 1e-2, 0.5)
    crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
    step = TrainStep(m, crit, arena, grad_clip=1.0, capture=capture)
    osd = {k: v.clone() for k, v in sd.items()}
    ost = {}
    for it in range(2):
        x, masks, edges = O.synthetic_batch(4, 128, seed=30 + it)
        ref_l, ref_norm, _ = O.train_step(osd, ost, x, masks, edges, base_lr=1e-3, wd=1e-2, enc_ratio=0.5, clip=1.0, cfg=cfg)
        got = step(x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda())
        assert abs(float(got["loss"]) - ref_l["loss"]) < 2e-3 * abs(ref_l["loss"]), (it, float(got["loss"]), ref_l)
        assert abs(float(arena.gnorm_sq.sqrt()) - ref_norm) < 2e-2 * ref_norm
    # Adam's first updates are ~lr*sign(g): elements whose gradient is rounding noise may flip sign, so compare
    # the fraction of elements whose update disagrees, not a max norm.
    st = m.state_dict()
    tot = bad = 0
    for k, v in osd.items():
        if O.is_buffer_key(k):
            continue
        upd_ref = (v.detach() - sd[k]).double()
        upd = (st[k].cpu() - sd[k]).double()
        thr = 0.2 * float(upd_ref.abs().max())
        if thr == 0:
            continue
        tot += upd.numel()
        bad += int(((upd - upd_ref).abs() > thr).sum())
    assert bad / tot < 0.01, f"{bad}/{tot} parameter elements updated differently from the oracle"
    for k in ("fusion.bn.running_mean", "decoder.decoder_blocks.1.bn1.running_var"):
        assert rel_err(st[k].float(), osd[k]) < 2e-3, k


def test_train_step_bf16_runs_and_decreases_loss():
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.utils.loss_functions import CODLoss
    m, sd, cfg = make_model("tiny", "bf16", train=True)
    arena = Arena(m)
    m.mark_params_changed()
    arena.set_hyper(2e-3, 1e-5, 1.0)
    crit = CODLoss().cuda()
    step = TrainStep(m, crit, arena, grad_clip=1.0)
    x, masks, edges = O.synthetic_batch(4, 64, seed=40)
    xs, ms, es = x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda()
    first = float(step(xs, ms, es)["loss"])
    for _ in range(15):
        last = float(step(xs, ms, es)["loss"])
    assert last == last and last < first, (first, last)


def test_inference_hipgraph_forward_matches_eager():
    """BASELINE config 3 shape of use: eval-mode bf16 forward captured once into a hipGraph and replayed."""
    m, sd, cfg = make_model("large", "bf16", seed=3)
    x = torch.randn(4, 3, 384, 384, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        ref = m(x)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(x)
        torch.cuda.current_stream().wait_stream(side)
        static_x = x.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(static_x)
        static_x.copy_(x)
        g.replay()
    torch.cuda.synchronize()
    for a, b in zip(out["predictions"] + [out["edge"]], ref["predictions"] + [ref["edge"]]):
        assert torch.equal(a, b), "graph replay must reproduce the eager forward bit for bit (no atomics in eval forward)"
    assert out["predictions"][2].shape == (4, 1, 384, 384)


def test_high_res_768_train_step_runs():
    """BASELINE config 5 shape: 768x768 bf16 train step (LDS-tiled PED upsample / multi-scale stress) on one GPU."""
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.utils.loss_functions import CODLoss
    m, sd, cfg = make_model("large", "bf16", seed=3, train=True)
    arena = Arena(m)
    m.mark_params_changed()
    arena.set_hyper(1e-4, 1e-5, 0.05)
    step = TrainStep(m, CODLoss().cuda(), arena, grad_clip=1.0)
    x, masks, edges = O.synthetic_batch(2, 768, seed=50)
    out = step(x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda())
    l0 = float(out["loss"])
    l1 = float(step(x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda())["loss"])
    assert l0 == l0 and l1 == l1 and float(arena.gnorm_sq) > 0
    m.eval()
    with torch.no_grad():
        o = m(x.cuda())
    assert o["predictions"][2].shape == (2, 1, 768, 768) and o["edge"].shape == (2, 1, 96, 96)


def test_segmented_graph_step_equals_eager_step():
    """The multi-GPU graph mode captures a hand-written backward in segments; on one rank it must update the parameters
    exactly like the autograd-driven eager step (up to float-atomic ordering)."""
    from spegnet_amd.engine.arena import Arena
    from spegnet_amd.engine.trainer import TrainStep
    from spegnet_amd.utils.loss_functions import CODLoss
    res = []
    for seg in (False, True):
        m, sd, cfg = make_model("tiny", "fp32", train=True)
        arena = Arena(m)
        m.mark_params_changed()
        arena.set_hyper(1e-3, 1e-2, 0.5)
        step = TrainStep(m, CODLoss().cuda(), arena, grad_clip=1.0, capture=seg, force_segmented=seg)
        losses, gnorms = [], []
        for it in range(2):   # Adam at lr 1e-3 amplifies float-atomic noise step by step; two steps stay comparable
            x, masks, edges = O.synthetic_batch(4, 128, seed=60 + it)
            losses.append(float(step(x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda())["loss"]))
            gnorms.append(float(arena.gnorm_sq))
        res.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, gnorms))
        if seg:
            plan = step._plan
            assert len(step.segments) == len(plan) == 4 and plan[-1][0] == 0 and plan[-1][2] == arena.size
            assert [p[2] for p in plan] == sorted(p[2] for p in plan)
    (l0, s0, g0), (l1, s1, g1) = res
    for a, b in zip(l0, l1):
        assert abs(a - b) < 2e-3 * abs(a), (l0, l1)
    # step 1 runs on identical parameters: the two backward implementations must agree up to float-atomic ordering.  Step 2 comes
    # after an Adam update (first step ~ lr * sign(g): near-zero gradients flip sign on that noise), so its norm is only loosely tied.
    assert abs(g0[0] - g1[0]) < 5e-3 * g0[0], (g0, g1)
    assert abs(g0[1] - g1[1]) < 0.5 * g0[1], (g0, g1)
    tot = bad = 0
    sd0 = O.init_state_dict(seed=3, cfg=O.HIERA_TINY_TEST)
    for k in s0:
        if not s0[k].is_floating_point() or O.is_buffer_key(k):
            continue
        u0, u1 = (s0[k].cpu() - sd0[k]).double(), (s1[k].cpu() - sd0[k]).double()
        thr = 0.25 * float(u0.abs().max())
        if thr == 0:
            continue
        tot += u0.numel(); bad += int(((u0 - u1).abs() > thr).sum())
    assert bad / tot < 0.02, f"{bad}/{tot} parameter elements differ between segmented-graph and eager training"
