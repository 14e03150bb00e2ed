"""Parameter inventory of SPEGNet with the reference's state_dict key names and shapes.

Key names are the drop-in boundary for checkpoints (reference: engine/predictor.py:277-278 load_state_dict;
head names from models/feature_integration.py:193-203,300-367 and models/object_detection.py:112-130,185-199,
282-307; trunk names are sam2's Hiera under `encoder.encoder.` from models/feature_encoding.py:159).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

HIERA_CONFIGS = {
    # variant -> (embed_dim, num_heads, stages, global_att_blocks, window_spec); channels per
    # models/feature_encoding.py:141-148, block structure per the sam2.1 hiera yaml files
    "tiny": dict(embed_dim=96, num_heads=1, stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9), window_spec=(8, 4, 14, 7)),
    "small": dict(embed_dim=96, num_heads=1, stages=(1, 2, 11, 2), global_att_blocks=(7, 10, 13), window_spec=(8, 4, 14, 7)),
    "base_plus": dict(embed_dim=112, num_heads=2, stages=(2, 3, 16, 3), global_att_blocks=(12, 16, 20), window_spec=(8, 4, 14, 7)),
    "large": dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8)),
    # structure-preserving miniature used by the fast tests (not a reference variant)
    "test_tiny": dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 2), global_att_blocks=(4,), window_spec=(8, 4, 4, 2)),
}
for _c in HIERA_CONFIGS.values():
    _c.update(q_pool=3, q_stride=2, bkg=(7, 7), mlp_ratio=4.0, ln_eps=1e-6)


def block_table(cfg) -> List[dict]:
    """(dim, dim_out, heads, window, q_stride, stage_end) per block, as sam2's Hiera.__init__ lays them out:
    the first block of stages 2-4 keeps the previous stage's window, pools q 2x2 and doubles dim and heads."""
    depth = sum(cfg["stages"])
    ends = [sum(cfg["stages"][: i + 1]) - 1 for i in range(len(cfg["stages"]))]
    pool_blocks = [e + 1 for e in ends[:-1]][: cfg["q_pool"]]
    dim, heads, stage = cfg["embed_dim"], cfg["num_heads"], 1
    out = []
    for i in range(depth):
        dim_out, window = dim, cfg["window_spec"][stage - 1]
        if i in cfg["global_att_blocks"]:
            window = 0
        if i - 1 in ends:
            dim_out, heads, stage = dim * 2, heads * 2, stage + 1
        out.append(dict(idx=i, dim=dim, dim_out=dim_out, heads=heads, window=window,
                        q_stride=cfg["q_stride"] if i in pool_blocks else 0, stage_end=i in ends))
        dim = dim_out
    return out


def param_specs(cfg) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(name, shape, kind)], kind in {weight, bias, ln_w, ln_b, bn_w, bn_b, rmean, rvar, nbt, pos}."""
    D = cfg["embed_dim"]
    e = "encoder.encoder."
    s: List[Tuple[str, Tuple[int, ...], str]] = [
        (e + "pos_embed", (1, D, *cfg["bkg"]), "pos"),
        (e + "pos_embed_window", (1, D, cfg["window_spec"][0], cfg["window_spec"][0]), "pos"),
        (e + "patch_embed.proj.weight", (D, 3, 7, 7), "weight"),
        (e + "patch_embed.proj.bias", (D,), "bias"),
    ]
    for b in block_table(cfg):
        p = f"{e}blocks.{b['idx']}."
        d, do = b["dim"], b["dim_out"]
        hid = int(do * cfg["mlp_ratio"])
        s += [(p + "norm1.weight", (d,), "ln_w"), (p + "norm1.bias", (d,), "ln_b"),
              (p + "attn.qkv.weight", (3 * do, d), "weight"), (p + "attn.qkv.bias", (3 * do,), "bias"),
              (p + "attn.proj.weight", (do, do), "weight"), (p + "attn.proj.bias", (do,), "bias"),
              (p + "norm2.weight", (do,), "ln_w"), (p + "norm2.bias", (do,), "ln_b"),
              (p + "mlp.layers.0.weight", (hid, do), "weight"), (p + "mlp.layers.0.bias", (hid,), "bias"),
              (p + "mlp.layers.1.weight", (do, hid), "weight"), (p + "mlp.layers.1.bias", (do,), "bias")]
        if d != do:
            s += [(p + "proj.weight", (do, d), "weight"), (p + "proj.bias", (do,), "bias")]

    def bn(p, c):
        return [(p + "weight", (c,), "bn_w"), (p + "bias", (c,), "bn_b"), (p + "running_mean", (c,), "rmean"),
                (p + "running_var", (c,), "rvar"), (p + "num_batches_tracked", (), "nbt")]

    tot = D * 2 + D * 4 + D * 8
    s += [("fusion.conv1x1.weight", (512, tot, 1, 1), "weight")] + bn("fusion.bn.", 512)
    s += [("fusion.se_block.fc.0.weight", (32, 512), "weight"), ("fusion.se_block.fc.2.weight", (512, 32), "weight")]
    s += [("context.reduce.0.weight", (128, 512, 1, 1), "weight")] + bn("context.reduce.1.", 128)
    for i in range(4):
        s += [(f"context.branches.{i}.0.weight", (128, 1, 3, 3), "weight")] + bn(f"context.branches.{i}.1.", 128)
    s += [("context.global_branch.1.weight", (128, 128, 1, 1), "weight")] + bn("context.global_branch.2.", 128)
    s += [("context.fusion.0.weight", (128, 5, 1, 1), "weight")] + bn("context.fusion.1.", 128)
    s += [("context.expand.0.weight", (256, 128, 1, 1), "weight")] + bn("context.expand.1.", 256)
    s += [("edge_detector.conv1.weight", (64, 256, 3, 3), "weight")] + bn("edge_detector.bn1.", 64)
    s += [("edge_detector.edge_conv.weight", (1, 64, 1, 1), "weight"), ("edge_detector.edge_conv.bias", (1,), "bias")]
    prev = 256
    for i, (c, ec) in enumerate(zip((256, 128, 64), (64, 64, 0))):
        p = f"decoder.decoder_blocks.{i}."
        s += [(p + "conv1.weight", (c, prev + ec, 3, 3), "weight"), (p + "conv1.bias", (c,), "bias")] + bn(p + "bn1.", c)
        s += [(p + "conv2.weight", (c, c, 3, 3), "weight"), (p + "conv2.bias", (c,), "bias")] + bn(p + "bn2.", c)
        prev = c
    for i, c in enumerate((256, 128, 64)):
        s += [(f"decoder.pred_heads.{i}.weight", (1, c, 1, 1), "weight"), (f"decoder.pred_heads.{i}.bias", (1,), "bias")]
    return s


BUFFER_KINDS = ("rmean", "rvar", "nbt")


def init_tensor(name: str, shape, kind: str, g: torch.Generator, empty: bool = False) -> torch.Tensor:
    """nn-default-style initialisation (head) / trunc-normal(0.02) (trunk); real use loads a checkpoint.  empty: the random tensors are left
    uninitialised (config['init'] = 'empty': the caller loads a state_dict next; drawing 215 M normals takes seconds)."""
    if empty and kind in ("pos", "weight", "bias"):
        return torch.empty(shape)
    if kind in ("ln_w", "bn_w", "rvar"):
        return torch.ones(shape)
    if kind in ("ln_b", "bn_b", "rmean"):
        return torch.zeros(shape)
    if kind == "nbt":
        return torch.zeros((), dtype=torch.long)
    trunk = name.startswith("encoder.")
    if kind == "pos":
        return torch.randn(shape, generator=g).clamp_(-2, 2) * 0.02
    if kind == "weight":
        if trunk and len(shape) == 2:
            return torch.randn(shape, generator=g).clamp_(-2, 2) * 0.02
        fan_in = int(torch.tensor(shape[1:]).prod()) if len(shape) > 1 else shape[0]
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b
    if kind == "bias":
        if trunk and "patch_embed" not in name:
            return torch.zeros(shape)
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    raise ValueError(kind)


class ParamTree(nn.Module):
    """A bare container whose parameter / buffer names reproduce a dotted key (digits become child names)."""

    def insert(self, key: str, tensor: torch.Tensor, is_buffer: bool):
        head, _, rest = key.partition(".")
        if not rest:
            if is_buffer:
                self.register_buffer(head, tensor)
            else:
                self.register_parameter(head, nn.Parameter(tensor))
            return
        if head not in self._modules:
            self.add_module(head, ParamTree())
        self._modules[head].insert(rest, tensor, is_buffer)


def build_tree(root: nn.Module, cfg, seed: int = 0) -> Dict[str, str]:
    """Populates `root` with every parameter / buffer of param_specs(cfg); returns {name: kind}."""
    g = torch.Generator().manual_seed(seed)
    kinds = {}
    for name, shape, kind in param_specs(cfg):
        top, _, rest = name.partition(".")
        if top not in root._modules:
            root.add_module(top, ParamTree())
        root._modules[top].insert(rest, init_tensor(name, shape, kind, g), kind in BUFFER_KINDS)
        kinds[name] = kind
    return kinds
