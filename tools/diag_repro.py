"""Run the same fp32 forward+backward several times; report run-to-run gradient differences (atomics ordering noise)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spegnet_oracle as O
from spegnet_amd.models import SPEGNet
from spegnet_amd.utils.loss_functions import CODLoss
cfg = O.HIERA_TINY_TEST
B, S = 4, 128
m = SPEGNet({"encoder": {"variant": "test_tiny"}, "compute_dtype": "fp32"})
m.load_state_dict(O.init_state_dict(seed=3, cfg=cfg)); m = m.cuda().train()
x, masks, edges = O.synthetic_batch(B, S, seed=20)
crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
runs = []
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
for r in range(4):
    m.load_state_dict(sd0)
    for p in m.parameters(): p.grad = None
    out = m(x.cuda())
    l = crit.forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
    l["loss"].backward()
    runs.append(({k: p.grad.clone() for k, p in m.named_parameters()}, float(l["loss"]), out["predictions"][2].detach().clone()))
gmax = max(float(g.abs().max()) for g in runs[0][0].values())
print("losses", [r[1] for r in runs])
print("fwd pred3 max diff run0-run1", float((runs[0][2] - runs[1][2]).abs().max()))
rows = []
for k in runs[0][0]:
    sc = max(float(runs[0][0][k].abs().max()), 1e-3 * gmax)
    d = max(float((runs[i][0][k] - runs[0][0][k]).abs().max()) for i in range(1, 4)) / sc
    rows.append((d, k))
rows.sort(reverse=True)
for d, k in rows[:12]:
    print(f"{d:.3e} {k}")
