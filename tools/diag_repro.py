"""Diagnostic: which gradients differ between two identical train-mode forward+backward passes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spegnet_oracle as O


def run(dtype, S, B, variant="large"):
    from spegnet_amd.models import SPEGNet
    from spegnet_amd.utils.loss_functions import CODLoss
    cfg = O.HIERA_L if variant == "large" else O.HIERA_TINY_TEST
    x, masks, edges = O.synthetic_batch(B, S, seed=21)
    xs, ms, es = x.cuda(), torch.stack(masks).cuda(), torch.stack(edges).cuda()
    gs = []
    for rep in range(3):
        sd = O.init_state_dict(seed=3, cfg=cfg)
        m = SPEGNet({"encoder": {"variant": variant if variant == "large" else "test_tiny"}, "compute_dtype": dtype})
        m.load_state_dict(sd)
        m = m.cuda().train()
        crit = CODLoss().cuda()
        out = m(xs)
        l = crit.forward_batched(out["predictions"], out["edge"], ms, es)
        l["loss"].backward()
        torch.cuda.synchronize()
        gs.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
    for r in (1, 2):
        bad = [(k, float((gs[0][k] - gs[r][k]).abs().max()), float(gs[0][k].abs().max())) for k in gs[0] if not torch.equal(gs[0][k], gs[r][k])]
        print(f"{dtype} S={S} B={B} run0 vs run{r}: {len(bad)} differing")
        for k, d, mx in bad:
            if not k.endswith(".bias"):
                print("   ", k, f"maxdiff {d:.3e} of {mx:.3e}")


if __name__ == "__main__":
    run("bf16", 128, 4)
    run("bf16", 256, 2)
    run("fp32", 128, 4)
