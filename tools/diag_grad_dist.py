"""Diagnostic (not a test): repeats the HIP fp32 train-mode backward N times on identical inputs and prints, per run, the percentiles
of the per-parameter gradient error against the fp64 oracle -- max-abs metric (what tests/test_model_gpu.py uses) and relative L2 --
plus the same for the fp32 oracle.  Shows how the float-atomic ordering noise of train-mode BatchNorm is distributed.
Usage: python tools/diag_grad_dist.py [runs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spegnet_oracle as O
import oracle.spegnet_oracle as OM
from spegnet_amd.models import SPEGNet
from spegnet_amd.utils.loss_functions import CODLoss

runs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
cfg, B, S = O.HIERA_TINY_TEST, 4, 128


def oracle(dtype):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in O.init_state_dict(seed=3, cfg=cfg).items()}
    OM._LAPLACE = OM._LAPLACE.to(dtype)
    x, masks, edges = O.synthetic_batch(B, S, seed=20)
    x = x.to(dtype); masks = [m.to(dtype) for m in masks]; edges = [e.to(dtype) for e in edges]
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not O.is_buffer_key(k)}
    out = O.spegnet_forward(sd, x, training=True, cfg=cfg)
    l = O.cod_loss(out['predictions'], out['edge'], masks, edges, **O.LOSS_DEFAULT_YAML)
    g = torch.autograd.grad(l['loss'], list(params.values()), allow_unused=True)
    OM._LAPLACE = OM._LAPLACE.float()
    return dict(zip(params.keys(), g))


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def stats(g, g64, gmax):
    emax, el2 = [], []
    for k, ref in g64.items():
        if ref is None:
            continue
        d = g[k].double().cpu() - ref
        emax.append(float(d.abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax))
        el2.append(float(d.norm()) / max(float(ref.norm()), 1e-30))
    return emax, el2


g64, g32 = oracle(torch.float64), oracle(torch.float32)
gmax = max(float(v.abs().max()) for v in g64.values() if v is not None)
em, el = stats(g32, g64, gmax)
fmt = lambda v: " ".join(f"p{int(q*100)}={pct(v, q):.1e}" for q in (0.5, 0.8, 0.95)) + f" max={max(v):.1e}"
print("fp32 oracle   max-abs:", fmt(em), "| rel-L2:", fmt(el))
x, masks, edges = O.synthetic_batch(B, S, seed=20)
crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
for r in range(runs):
    m = SPEGNet({"encoder": {"variant": "test_tiny"}, "compute_dtype": "fp32"})
    m.load_state_dict(O.init_state_dict(seed=3, cfg=cfg))
    m = m.cuda().train()
    out = m(x.cuda())
    l = crit.forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
    l["loss"].backward()
    g = {k: p.grad for k, p in m.named_parameters()}
    em, el = stats(g, g64, gmax)
    print(f"hip run {r}     max-abs:", fmt(em), "| rel-L2:", fmt(el), flush=True)
    if pct(em, 0.5) > 3e-3 and "--detail" in sys.argv:
        keys = [k for k, ref in g64.items() if ref is not None]
        for k, e in sorted(zip(keys, em), key=lambda kv: -kv[1]):
            if not k.startswith("encoder."):
                print(f"      {e:.2e}  {k}")
        sys.exit(0)
