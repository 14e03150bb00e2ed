"""Generates the golden vectors under tests/golden/*.pt.  Run ONLY in the build container:

    python tests/golden/make_golden.py

It imports the reference's own importable modules from /root/reference (never copied into this repo)
-- models/feature_integration.py, models/object_detection.py, utils/loss_functions.py, utils/image_processor.py -- runs them on
seeded inputs with a seeded state_dict (oracle.init_state_dict, regenerated at test time, so only the
OUTPUTS are stored) and records what they return.  For the Hiera trunk, whose code lives in the absent
third-party `sam2` package, it records the outputs of transformers' independent Sam2HieraDetModel
(secondary cross-check; see oracle header).  The fixtures are data only.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import spegnet_oracle as O  # noqa: E402


def _ref_head():
    sys.path.insert(0, REF)
    from models.feature_integration import AdaptiveAttentionFusion, EfficientASPP
    from models.object_detection import EdgeDetectionModule, BoundaryAwareDecoder
    sys.path.pop(0)
    return torch.nn.ModuleDict(dict(
        fusion=AdaptiveAttentionFusion([288, 576, 1152], 512),
        context=EfficientASPP(512, 256, 4, [1, 6, 12, 18]),
        edge_detector=EdgeDetectionModule(256, 64),
        decoder=BoundaryAwareDecoder(256, [256, 128, 64], 1, [64, 64, None])))


def _ref_forward(m, feats):
    # same wiring as models/spegnet.py:169-206
    fused = m["fusion"](feats)
    context = m["context"](fused)
    edge_map, ef = m["edge_detector"](context)
    preds = m["decoder"](context, edge_features_list=[ef, ef, None])
    return dict(predictions=preds, edge=edge_map, fused=fused, context=context, edge_features=ef)


def head_inputs(B, h, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, c, h // d, h // d, generator=g) for c, d in ((288, 1), (576, 2), (1152, 4))]


def sd_checksum(sd):
    return float(sum(v.double().abs().sum() for k, v in sd.items() if v.is_floating_point()))


def gen_head():
    out = {}
    for tag, (B, h, training) in {"eval_small": (2, 8, False), "train_small": (3, 8, True),
                                  "eval_384": (1, 48, False)}.items():
        sd = {k: v for k, v in O.init_state_dict(seed=11).items() if not k.startswith("encoder.")}
        m = _ref_head()
        m.load_state_dict(sd, strict=True)
        m.train(training)
        feats = [f.requires_grad_(True) for f in head_inputs(B, h, seed=5)]
        r = _ref_forward(m, feats)
        gw = torch.Generator().manual_seed(7)
        ws = [torch.randn(p.shape, generator=gw) for p in r["predictions"]] + [torch.randn(r["edge"].shape, generator=gw)]
        scalar = sum((w * p).sum() for w, p in zip(ws, r["predictions"] + [r["edge"]]))
        rec = {"B": B, "h": h, "training": training, "sd_seed": 11, "in_seed": 5, "w_seed": 7,
               "sd_checksum": sd_checksum(sd)}
        if tag == "eval_384":
            rec["pred1"] = r["predictions"][0].detach()
            rec["pred2_s2"] = r["predictions"][1].detach()[..., ::2, ::2].contiguous()
            rec["pred3_s4"] = r["predictions"][2].detach()[..., ::4, ::4].contiguous()
            rec["edge"] = r["edge"].detach()
            for k in ("fused", "context", "edge_features"):
                rec[k + "_chanmean"] = r[k].detach().mean((2, 3))
        else:
            scalar.backward()
            rec["predictions"] = [p.detach() for p in r["predictions"]]
            rec["edge"] = r["edge"].detach()
            for k in ("fused", "context", "edge_features"):
                rec[k] = r[k].detach()
            rec["grad_inputs"] = [f.grad.clone() for f in feats]
            rec["grad_norms"] = {k: float(p.grad.norm()) for k, p in m.named_parameters()}
            keep = ("fusion.bn.weight", "fusion.se_block.fc.0.weight", "context.fusion.0.weight",
                    "context.branches.2.0.weight", "context.global_branch.2.bias", "edge_detector.edge_conv.weight",
                    "decoder.pred_heads.1.weight", "decoder.decoder_blocks.2.conv2.bias",
                    "decoder.decoder_blocks.2.bn1.weight", "context.expand.1.bias")
            rec["grads"] = {k: p.grad.clone() for k, p in m.named_parameters() if k in keep}
            if training:
                st = m.state_dict()
                rec["running"] = {k: st[k].clone() for k in st if k.endswith(("running_mean", "running_var"))
                                  and k.split(".")[0] in ("fusion", "edge_detector") or k.startswith("context.global_branch.2.running")}
        out[tag] = rec
    torch.save(out, os.path.join(HERE, "head.pt"))
    print("head.pt", {k: list(v.keys())[:4] for k, v in out.items()})


def loss_case(name):
    g = torch.Generator().manual_seed({"rand": 21, "zeros": 22, "ones": 23, "ragged": 24}[name])
    B, S = 2, 64
    sizes = [(S, S)] * B if name != "ragged" else [(50, 70), (96, 64)]
    preds = [torch.randn(B, 1, S // d, S // d, generator=g) * 2 for d in (4, 2, 1)]
    edge = torch.randn(B, 1, S // 8, S // 8, generator=g) * 2
    masks, edges = [], []
    for (hh, ww) in sizes:
        if name == "zeros":
            masks.append(torch.zeros(1, hh, ww)); edges.append(torch.zeros(1, hh, ww))
        elif name == "ones":
            masks.append(torch.ones(1, hh, ww)); edges.append(torch.ones(1, hh, ww))
        else:
            masks.append((torch.rand(1, hh, ww, generator=g) > 0.7).float())
            edges.append((torch.rand(1, hh, ww, generator=g) > 0.95).float())
    return preds, edge, masks, edges


def gen_loss():
    sys.path.insert(0, REF)
    from utils.loss_functions import CODLoss
    import torch.nn.functional as F
    sys.path.pop(0)
    out = {}
    for cfg_name, cfg in {"yaml": dict(O.LOSS_DEFAULT_YAML), "ctor_default": {}}.items():
        c = dict(cfg)
        if "scale_weights" in c:
            c["scale_weights"] = list(c["scale_weights"])
        crit = CODLoss(**c)
        for name in ("rand", "zeros", "ones", "ragged"):
            preds, edge, masks, edges = loss_case(name)
            preds = [p.requires_grad_(True) for p in preds]
            edge = edge.requires_grad_(True)
            # resize loop of engine/trainer.py:358-383
            bp, be = [], []
            for i in range(len(masks)):
                bp.append([F.interpolate(p[i:i + 1], size=masks[i].shape[-2:], mode="bilinear", align_corners=False) for p in preds])
                be.append(F.interpolate(edge[i:i + 1], size=edges[i].shape[-2:], mode="bilinear", align_corners=False))
            ld = crit(predictions=bp, edge_pred=be, masks=masks, edges=edges)
            ld["loss"].backward()
            out[f"{cfg_name}/{name}"] = {"loss": {k: float(v) for k, v in ld.items()},
                                         "grad_preds": [p.grad.clone() for p in preds], "grad_edge": edge.grad.clone()}
            wm = crit.compute_boundary_weights([masks[0]])[0]
            out[f"{cfg_name}/{name}"]["wmap0_sum"] = float(wm.double().sum())
            out[f"{cfg_name}/{name}"]["wmap0_corner"] = wm[:, :6, :6].clone()
    torch.save(out, os.path.join(HERE, "loss.pt"))
    print("loss.pt", {k: v["loss"] for k, v in out.items()})


HF_KEYMAP = (("patch_embed.proj.", "patch_embed.projection."), (".norm1.", ".layer_norm1."), (".norm2.", ".layer_norm2."),
             (".mlp.layers.0.", ".mlp.proj_in."), (".mlp.layers.1.", ".mlp.proj_out."))


def gen_trunk():
    """Secondary cross-check: transformers' Sam2HieraDetModel with OUR seeded weights."""
    from transformers.models.sam2.configuration_sam2 import Sam2HieraDetConfig
    from transformers.models.sam2.modeling_sam2 import Sam2HieraDetModel
    out = {}
    for tag, (cfg, S, B) in {"tiny_64": (O.HIERA_TINY_TEST, 64, 2), "tiny_96": (O.HIERA_TINY_TEST, 96, 1),
                             "large_64": (O.HIERA_L, 64, 1), "large_128": (O.HIERA_L, 128, 1),
                             # 384 px: the BASELINE resolution -- stage 3 pads 24 -> 32 (window 16), stage 4 pads 12 -> 16 (window 8)
                             "large_384": (O.HIERA_L, 384, 1)}.items():
        dims = [cfg["embed_dim"] * 2 ** i for i in range(4)]
        heads = [cfg["num_heads"] * 2 ** i for i in range(4)]
        hc = Sam2HieraDetConfig(hidden_size=cfg["embed_dim"], num_attention_heads=cfg["num_heads"],
                                blocks_per_stage=list(cfg["stages"]), embed_dim_per_stage=dims,
                                num_attention_heads_per_stage=heads, window_size_per_stage=list(cfg["window_spec"]),
                                global_attention_blocks=list(cfg["global_att_blocks"]), num_query_pool_stages=cfg["q_pool"],
                                window_positional_embedding_background_size=list(cfg["bkg"]), image_size=[S, S])
        hc._attn_implementation = "eager"
        m = Sam2HieraDetModel(hc).eval()
        sd = {k[len("encoder.encoder."):]: v for k, v in O.init_state_dict(seed=3, cfg=cfg).items() if k.startswith("encoder.encoder.")}
        hsd = {}
        for k, v in sd.items():
            for a, b in HF_KEYMAP:
                k = k.replace(a, b)
            hsd[k] = v
        missing = m.load_state_dict(hsd, strict=True)
        x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            r = m(pixel_values=x)
        feats = [f.permute(0, 3, 1, 2).contiguous() for f in r.intermediate_hidden_states]
        out[tag] = {"S": S, "B": B, "sd_seed": 3, "in_seed": 9, "feats": feats if "tiny" in tag else None,
                    "feat_stats": [(float(f.mean()), float(f.std()), float(f.abs().max())) for f in feats],
                    "feat_s4": feats[3], "feat_s3_slice": feats[2][:, ::8].contiguous(), "feat_s2_slice": feats[1][:, ::16].contiguous()}
        print(tag, [tuple(f.shape) for f in feats], str(missing))
    torch.save(out, os.path.join(HERE, "trunk_hf.pt"))


def gen_preprocess():
    """The reference's own CODImageProcessor.process_image (utils/image_processor.py:94-134) on two small synthetic PNGs written to a
    temporary directory: stores the decoded uint8 pixels (the INPUT of the arithmetic) and the tensor the reference returns."""
    import tempfile
    import numpy as np
    from PIL import Image
    sys.path.insert(0, REF)
    from utils.image_processor import CODImageProcessor
    sys.path.pop(0)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, (H, W, S) in {"down": (97, 131, 64), "up": (40, 56, 64)}.items():
            g = torch.Generator().manual_seed(H + W)
            px = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
            path = os.path.join(d, name + ".png")
            Image.fromarray(px.numpy()).save(path)
            proc = CODImageProcessor(target_size=S)
            out[name] = {"pixels": px, "size": S, "out": proc.process_image(path).clone()}
    torch.save(out, os.path.join(HERE, "preprocess.pt"))
    print("preprocess.pt", {k: tuple(v["out"].shape) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["head", "loss", "trunk", "preprocess"]
    if "head" in which:
        gen_head()
    if "loss" in which:
        gen_loss()
    if "trunk" in which:
        gen_trunk()
    if "preprocess" in which:
        gen_preprocess()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
