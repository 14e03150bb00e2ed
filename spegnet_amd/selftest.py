"""smoke(): one tiny forward+backward of the HIP SPEGNet path on cuda:0, checked against the CPU oracle."""
import os
import sys

import torch


def smoke() -> None:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import spegnet_oracle as O   # checker only
    from spegnet_amd.models import SPEGNet
    from spegnet_amd.utils.loss_functions import CODLoss
    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    cfg = O.HIERA_TINY_TEST
    sd = O.init_state_dict(seed=1, cfg=cfg)
    m = SPEGNet({"encoder": {"variant": "test_tiny"}, "compute_dtype": "fp32"})
    m.load_state_dict(sd)
    m = m.to("cuda:0").train()
    x, masks, edges = O.synthetic_batch(2, 64, seed=5)
    ref = O.spegnet_forward({k: v.clone() for k, v in sd.items()}, x, training=True, cfg=cfg)
    out = m(x.cuda())
    err = float((out["predictions"][2].float().cpu() - ref["predictions"][2]).abs().max() / ref["predictions"][2].abs().max())
    assert err < 1e-3, f"smoke forward mismatch vs oracle: {err}"
    loss = CODLoss().cuda().forward_batched(out["predictions"], out["edge"], torch.stack(masks).cuda(), torch.stack(edges).cuda())
    loss["loss"].backward()
    g = m.encoder.encoder.blocks[0].attn.qkv.weight.grad if hasattr(m.encoder.encoder.blocks, "__getitem__") else None
    gn = sum(float(p.grad.float().norm()) for p in m.parameters() if p.grad is not None)
    assert gn == gn and gn > 0, "smoke backward produced no gradient"
    torch.cuda.synchronize()
    print(f"smoke ok: fwd rel err {err:.2e}, loss {float(loss['loss']):.4f}, sum|grad| {gn:.3e}")
