"""Diagnostic (not a test): per-parameter gradient error of the HIP fp32 path and of the fp32 oracle, both against
the fp64 oracle.  Usage: python tools/diag_grads.py [variant] [B] [S]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spegnet_oracle as O
import oracle.spegnet_oracle as OM
from spegnet_amd.models import SPEGNet
from spegnet_amd.utils.loss_functions import CODLoss

variant = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = O.HIERA_L if variant == "large" else O.HIERA_TINY_TEST


def oracle(dtype):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in O.init_state_dict(seed=3, cfg=cfg).items()}
    OM._LAPLACE = OM._LAPLACE.to(dtype)
    x, masks, edges = O.synthetic_batch(B, S, seed=20)
    x = x.to(dtype); masks = [m.to(dtype) for m in masks]; edges = [e.to(dtype) for e in edges]
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not O.is_buffer_key(k)}
    out = O.spegnet_forward(sd, x, training=True, cfg=cfg)
    l = O.cod_loss(out['predictions'], out['edge'], masks, edges, **O.LOSS_DEFAULT_YAML)
    g = torch.autograd.grad(l['loss'], list(params.values()), allow_unused=True)
    return dict(zip(params.keys(), g)), float(l['loss'])


g64, l64 = oracle(torch.float64)
g32, l32 = oracle(torch.float32)
m = SPEGNet({"encoder": {"variant": "large" if variant == "large" else "test_tiny"}, "compute_dtype": "fp32"})
m.load_state_dict(O.init_state_dict(seed=3, cfg=cfg))
m = m.cuda().train()
x, masks, edges = O.synthetic_batch(B, S, seed=20)
crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
out = m(x.cuda())
l = crit.forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
l["loss"].backward()
print("loss fp64 %.8f fp32-oracle %.8f hip %.8f" % (l64, l32, float(l["loss"])))
gmax = max(float(v.abs().max()) for v in g64.values() if v is not None)
rows = []
for k, p in m.named_parameters():
    if g64[k] is None:
        continue
    scale = max(float(g64[k].abs().max()), 1e-3 * gmax)
    e_hip = float((p.grad.cpu().double() - g64[k]).abs().max()) / scale
    e_o32 = float((g32[k].double() - g64[k]).abs().max()) / scale
    rows.append((e_hip, e_o32, k, float(g64[k].abs().max())))
rows.sort(reverse=True)
for r in rows[:40]:
    print("hip %.2e  oracle32 %.2e  |g|max %.2e  %s" % (r[0], r[1], r[3], r[2]))
print("median hip %.2e oracle32 %.2e" % (sorted(r[0] for r in rows)[len(rows)//2], sorted(r[1] for r in rows)[len(rows)//2]))
