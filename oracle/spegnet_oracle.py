"""CPU oracle for the SPEGNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product path (``spegnet_amd``) never imports it and has no CPU fallback.

What it is: a plain-PyTorch fp32 *functional* restatement (state-dict in, tensors out) of the
algorithm the reference runs on its hot path.  Every function cites the reference lines it follows
(paths relative to the reference checkout).

Pinning status
  * head (CFI / EFE / PED), CODLoss: PINNED -- checked bit-for-bit/1e-6 against golden vectors that
    ``tests/golden/make_golden.py`` produced by importing the reference's own modules
    (``models/feature_integration.py``, ``models/object_detection.py``, ``utils/loss_functions.py``).
  * Hiera-L trunk: the reference takes it from the un-vendored, un-pinned third-party package
    ``sam2`` (setup/environment.yml:25, models/feature_encoding.py:107,156-159) which is absent
    here => "parity unpinned" by the reference.  The restatement follows the published Hiera
    algorithm (facebookresearch/sam2 ``modeling/backbones/hieradet.py``) and is cross-checked against
    the independent re-implementation shipped in ``transformers`` 5.15 (``Sam2HieraDetModel``) through
    golden vectors generated from it (secondary pin, see tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------------------------
# Hiera-L trunk (reference: models/feature_encoding.py:156-159,236 -> sam2 Hiera)
# --------------------------------------------------------------------------------------------
HIERA_L = dict(
    embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), q_pool=3, q_stride=2,
    window_spec=(8, 4, 16, 8), global_att_blocks=(23, 33, 43), bkg=(7, 7), mlp_ratio=4.0,
    ln_eps=1e-6,
)
# tiny configuration with the same structure (used by fast tests)
HIERA_TINY_TEST = dict(
    embed_dim=16, num_heads=1, stages=(1, 2, 3, 2), q_pool=3, q_stride=2,
    window_spec=(8, 4, 4, 2), global_att_blocks=(4,), bkg=(7, 7), mlp_ratio=4.0, ln_eps=1e-6,
)


def hiera_block_table(cfg=HIERA_L) -> List[dict]:
    """Per-block (dim, dim_out, heads, window, q_stride) exactly as sam2's Hiera.__init__ derives
    them: the first block of stages 2..4 keeps the previous stage's window ("lags by a block"),
    pools q by 2x2 and doubles dim/heads; global blocks use window 0."""
    depth = sum(cfg["stages"])
    stage_ends = [sum(cfg["stages"][: i + 1]) - 1 for i in range(len(cfg["stages"]))]
    q_pool_blocks = [e + 1 for e in stage_ends[:-1]][: cfg["q_pool"]]
    dim, heads, cur_stage = cfg["embed_dim"], cfg["num_heads"], 1
    table = []
    for i in range(depth):
        dim_out = dim
        window = cfg["window_spec"][cur_stage - 1]
        if i in cfg["global_att_blocks"]:
            window = 0
        if i - 1 in stage_ends:
            dim_out = dim * 2
            heads = heads * 2
            cur_stage += 1
        table.append(dict(idx=i, dim=dim, dim_out=dim_out, heads=heads, window=window,
                          q_stride=cfg["q_stride"] if i in q_pool_blocks else 0,
                          stage_end=i in stage_ends))
        dim = dim_out
    return table


def _window_partition(x: Tensor, w: int) -> Tuple[Tensor, Tuple[int, int]]:
    B, H, W, C = x.shape
    ph, pw = (-H) % w, (-W) % w
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // w, w, Wp // w, w, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, w, w, C), (Hp, Wp)


def _window_unpartition(x: Tensor, w: int, pad_hw: Tuple[int, int], hw: Tuple[int, int]) -> Tensor:
    Hp, Wp = pad_hw
    H, W = hw
    B = x.shape[0] // ((Hp // w) * (Wp // w))
    x = x.view(B, Hp // w, Wp // w, w, w, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def _pool2(x: Tensor, s: int) -> Tensor:  # x: [B,H,W,C]
    return F.max_pool2d(x.permute(0, 3, 1, 2), s, s).permute(0, 2, 3, 1)


def hiera_pos_embed(sd: SD, pre: str, hw: Tuple[int, int]) -> Tensor:
    """bicubic(pos_embed -> hw) + tiled window embedding, returned NHWC [1,h,w,C]."""
    pe = F.interpolate(sd[pre + "pos_embed"], size=hw, mode="bicubic")
    we = sd[pre + "pos_embed_window"]
    pe = pe + we.tile([a // b for a, b in zip(pe.shape, we.shape)])
    return pe.permute(0, 2, 3, 1)


def hiera_block(sd: SD, pre: str, blk: dict, x: Tensor, eps: float) -> Tensor:
    """One MultiScaleBlock.  x: [B,H,W,dim] -> [B,H',W',dim_out]."""
    p = f"{pre}blocks.{blk['idx']}."
    dim, dim_out, nh, window, qs = blk["dim"], blk["dim_out"], blk["heads"], blk["window"], blk["q_stride"]
    shortcut = x
    x = F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    if dim != dim_out:
        shortcut = _pool2(F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"]), qs)
    H, W = x.shape[1], x.shape[2]
    pad_hw = (H, W)
    if window > 0:
        x, pad_hw = _window_partition(x, window)
    # attention (q pooled inside each window)
    Bw, h, w, _ = x.shape
    qkv = F.linear(x, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(Bw, h * w, 3, nh, -1)
    q, k, v = qkv.unbind(2)
    if qs:
        q = _pool2(q.reshape(Bw, h, w, -1), qs)
        h, w = q.shape[1], q.shape[2]
        q = q.reshape(Bw, h * w, nh, -1)
    hd = q.shape[-1]
    att = (q.transpose(1, 2) * hd ** -0.5) @ k.transpose(1, 2).transpose(-2, -1)
    att = att.softmax(dim=-1)
    o = (att @ v.transpose(1, 2)).transpose(1, 2).reshape(Bw, h, w, -1)
    x = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    win_out = window
    if qs:
        win_out = window // qs
        H, W = shortcut.shape[1], shortcut.shape[2]
        pad_hw = (H + (-H) % win_out if win_out else H, W + (-W) % win_out if win_out else W)
    if window > 0:
        x = _window_unpartition(x, win_out, pad_hw, (H, W))
    x = shortcut + x
    y = F.layer_norm(x, (dim_out,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    y = F.linear(y, sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + "mlp.layers.1.weight"], sd[p + "mlp.layers.1.bias"])
    return x + y


def hiera_trunk(sd: SD, x: Tensor, pre: str = "encoder.encoder.", cfg=HIERA_L) -> List[Tensor]:
    """x [B,3,S,S] -> 4 NCHW stage maps (models/feature_encoding.py:209-236)."""
    if x.dim() != 4:
        raise ValueError(f"Expected 4D input (B,C,H,W), got {x.dim()}D")
    if any(s % 32 != 0 for s in x.shape[-2:]):
        raise ValueError("Input spatial dims must be divisible by 32")
    x = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, pre, (x.shape[1], x.shape[2]))
    outs = []
    for blk in hiera_block_table(cfg):
        x = hiera_block(sd, pre, blk, x, cfg["ln_eps"])
        if blk["stage_end"]:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


# --------------------------------------------------------------------------------------------
# Head: CFI / EFE / PED
# --------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1); train mode updates running stats in sd."""
    if training and p + "num_batches_tracked" in sd:
        sd[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training, 0.1, 1e-5)


def _bilinear(x: Tensor, size) -> Tensor:
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def se_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """models/feature_integration.py:128-151."""
    y = x.mean((2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, sd[p + "fc.0.weight"])), sd[p + "fc.2.weight"]))
    return x * y[:, :, None, None]


def cfi_fusion(sd: SD, feats: Sequence[Tensor], training: bool, p: str = "fusion.") -> Tensor:
    """AdaptiveAttentionFusion.forward, models/feature_integration.py:205-246."""
    size = feats[0].shape[2:]
    x = torch.cat([f if f.shape[2:] == size else _bilinear(f, size) for f in feats], 1)
    x = F.conv2d(x, sd[p + "conv1x1.weight"])
    x = F.relu(_bn(sd, p + "bn.", x, training))
    return se_block(sd, p + "se_block.", x)


EASPP_RATES = (1, 6, 12, 18)


def cfi_easpp(sd: SD, x: Tensor, training: bool, p: str = "context.") -> Tensor:
    """EfficientASPP.forward, models/feature_integration.py:369-417."""
    size = x.shape[2:]
    x = F.relu(_bn(sd, p + "reduce.1.", F.conv2d(x, sd[p + "reduce.0.weight"]), training))
    C = x.shape[1]
    outs = []
    for i, r in enumerate(EASPP_RATES):
        y = F.conv2d(x, sd[f"{p}branches.{i}.0.weight"], padding=r, dilation=r, groups=C)
        outs.append(F.relu(_bn(sd, f"{p}branches.{i}.1.", y, training)))
    g = F.conv2d(x.mean((2, 3), keepdim=True), sd[p + "global_branch.1.weight"])
    g = F.relu(_bn(sd, p + "global_branch.2.", g, training))
    outs.append(g.expand(-1, -1, *size))  # bilinear 1x1 -> HxW is a broadcast
    x = torch.cat(outs, 1)
    # grouped 1x1: group g reads concat channels 5g..5g+4 (branch-major concat!)
    x = F.relu(_bn(sd, p + "fusion.1.", F.conv2d(x, sd[p + "fusion.0.weight"], groups=C), training))
    return F.relu(_bn(sd, p + "expand.1.", F.conv2d(x, sd[p + "expand.0.weight"]), training))


def efe(sd: SD, x: Tensor, training: bool, p: str = "edge_detector.") -> Tuple[Tensor, Tensor]:
    """EdgeDetectionModule.forward, models/object_detection.py:132-157."""
    f = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], padding=1), training))
    return F.conv2d(f, sd[p + "edge_conv.weight"], sd[p + "edge_conv.bias"]), f


def ped_block(sd: SD, p: str, x: Tensor, edge: Optional[Tensor], training: bool) -> Tensor:
    """DecoderBlock.forward, models/object_detection.py:201-238."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    if edge is not None:
        x = torch.cat([x, _bilinear(edge, x.shape[2:])], 1)
    x = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1), training))
    x = F.relu(_bn(sd, p + "bn2.", F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), training))
    return x


def ped(sd: SD, x: Tensor, edges: Sequence[Optional[Tensor]], training: bool, p: str = "decoder.") -> List[Tensor]:
    """BoundaryAwareDecoder.forward, models/object_detection.py:309-342."""
    preds = []
    for i, e in enumerate(edges):
        x = ped_block(sd, f"{p}decoder_blocks.{i}.", x, e, training)
        preds.append(F.conv2d(x, sd[f"{p}pred_heads.{i}.weight"], sd[f"{p}pred_heads.{i}.bias"]))
    return preds


def head_forward(sd: SD, feats: Sequence[Tensor], training: bool = False) -> dict:
    """SPEGNet.forward after the encoder, models/spegnet.py:169-206.  feats = [s2, s3, s4]."""
    fused = cfi_fusion(sd, feats, training)
    context = cfi_easpp(sd, fused, training)
    edge_map, edge_f = efe(sd, context, training)
    preds = ped(sd, context, [edge_f, edge_f, None], training)
    return {"predictions": preds, "edge": edge_map,
            "features": {"context": context, "fused": fused, "edge_features": edge_f}}


def spegnet_forward(sd: SD, x: Tensor, training: bool = False, cfg=HIERA_L) -> dict:
    """SPEGNet.forward, models/spegnet.py:137-206 (stage-1 map computed but unused, :169-171)."""
    feats = hiera_trunk(sd, x, cfg=cfg)
    return head_forward(sd, feats[1:4], training)


# --------------------------------------------------------------------------------------------
# CODLoss (utils/loss_functions.py:70-295; config values configs/default.yaml:34-43)
# --------------------------------------------------------------------------------------------
LOSS_DEFAULT_YAML = dict(scale_weights=(0.2, 0.3, 0.5), boundary_weight=2.0, bce_weight=1.25, iou_weight=1.0,
                         edge_weight=0.75, edge_focal_alpha=0.75, edge_focal_gamma=2.0)

_LAPLACE = torch.tensor([[-1., -1., -1.], [-1., 8., -1.], [-1., -1., -1.]]).view(1, 1, 3, 3)


def boundary_weights(mask: Tensor, boundary_weight: float) -> Tensor:
    """mask [1,H,W] -> weight map [1,H,W] (loss_functions.py:114-148)."""
    m = mask[None]
    lap = F.conv2d(m, _LAPLACE.to(m), padding=1).abs()
    dist = (F.avg_pool2d(m, 31, 1, 15) - m).abs()  # count_include_pad=True (default)
    return (1.0 + boundary_weight * (lap + dist))[0]


def structure_loss(pred: Tensor, mask: Tensor, wmap: Tensor, bce_w: float, iou_w: float) -> Tensor:
    """pred [1,1,H,W] logits; mask,wmap [1,H,W] (loss_functions.py:150-199)."""
    m, w = mask[None], wmap[None]
    npos, nneg = m.sum((2, 3), keepdim=True), (1 - m).sum((2, 3), keepdim=True)
    pw = (nneg / (npos + 1e-7)).clamp(0.1, 10.0)
    bce = F.binary_cross_entropy_with_logits(pred, m, pos_weight=pw, reduction="none")
    wbce = (w * bce).sum((2, 3)) / w.sum((2, 3))
    s = torch.sigmoid(pred)
    inter = (s * m * w).sum((2, 3))
    union = ((s + m) * w).sum((2, 3))
    wiou = 1 - (inter + 1) / (union - inter + 1)
    return (bce_w * wbce + iou_w * wiou).mean()


def edge_loss(pred: Tensor, target: Tensor, alpha: float, gamma: float) -> Tensor:
    """pred,target [1,1,H,W] (loss_functions.py:201-240)."""
    s = torch.sigmoid(pred)
    npos, nneg = target.sum((2, 3), keepdim=True), (1 - target).sum((2, 3), keepdim=True)
    pw = (nneg / (npos + 1e-7)).clamp(0.1, 10.0)
    pt = target * s + (1 - target) * (1 - s)
    focal = -pw * alpha * (1 - pt).pow(gamma) * torch.log(pt.clamp(min=1e-7))
    inter = (s * target).sum((2, 3))
    union = s.sum((2, 3)) + target.sum((2, 3))
    dice = 1 - (2 * inter + 1) / (union + 1)
    return focal.mean() + dice.mean()


def cod_loss(predictions: Sequence[Tensor], edge: Tensor, masks: Sequence[Tensor], edges: Sequence[Tensor],
             scale_weights=(0.2, 0.3, 0.5), boundary_weight=5.0, bce_weight=0.4, iou_weight=0.6,
             edge_weight=0.75, edge_focal_alpha=0.75, edge_focal_gamma=2.0) -> Dict[str, Tensor]:
    """Trainer._process_batch's resize loop (engine/trainer.py:358-383) + CODLoss.forward
    (utils/loss_functions.py:242-295).  predictions: 3 tensors [B,1,h,w]; edge [B,1,h,w];
    masks/edges: lists of [1,Hi,Wi]."""
    B = len(masks)
    seg_tot, edge_tot = 0.0, 0.0
    for i in range(B):
        wmap = boundary_weights(masks[i], boundary_weight)
        seg = 0.0
        for p, sw in zip(predictions, scale_weights):
            pi = _bilinear(p[i:i + 1], masks[i].shape[-2:])
            seg = seg + sw * structure_loss(pi, masks[i], wmap, bce_weight, iou_weight)
        ei = _bilinear(edge[i:i + 1], edges[i].shape[-2:])
        edge_tot = edge_tot + edge_loss(ei, edges[i][None], edge_focal_alpha, edge_focal_gamma)
        seg_tot = seg_tot + seg
    seg_avg, edge_avg = seg_tot / B, edge_tot / B
    return {"loss": seg_avg + edge_weight * edge_avg, "seg_loss": seg_avg, "edge_loss": edge_avg}


# --------------------------------------------------------------------------------------------
# Parameter construction (random init of the reference architecture; there is no checkpoint)
# --------------------------------------------------------------------------------------------
def init_state_dict(seed: int = 0, cfg=HIERA_L, dtype=torch.float32) -> SD:
    """Seeded random state_dict with the reference's key names and shapes.
    Head: nn-default-like kaiming-uniform; trunk: trunc_normal(0.02) weights, zero biases, unit LN.
    BN running stats are randomised a little so eval-mode tests are not trivial."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * b

    def tn(shape, std=0.02):
        return torch.randn(shape, generator=g, dtype=dtype).clamp_(-2, 2) * std

    def bn(p, c):
        sd[p + "weight"] = 1.0 + 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "bias"] = 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "running_mean"] = 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g, dtype=dtype)
        sd[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    e = "encoder.encoder."
    D = cfg["embed_dim"]
    sd[e + "patch_embed.proj.weight"] = uni((D, 3, 7, 7), 147)
    sd[e + "patch_embed.proj.bias"] = uni((D,), 147)
    sd[e + "pos_embed"] = tn((1, D, *cfg["bkg"]))
    sd[e + "pos_embed_window"] = tn((1, D, cfg["window_spec"][0], cfg["window_spec"][0]))
    for blk in hiera_block_table(cfg):
        p = f"{e}blocks.{blk['idx']}."
        d, do = blk["dim"], blk["dim_out"]
        for n, c in (("norm1", d), ("norm2", do)):
            sd[p + n + ".weight"] = 1.0 + 0.05 * torch.randn(c, generator=g, dtype=dtype)
            sd[p + n + ".bias"] = 0.02 * torch.randn(c, generator=g, dtype=dtype)
        hid = int(do * cfg["mlp_ratio"])
        for n, (o, i) in (("attn.qkv", (3 * do, d)), ("attn.proj", (do, do)),
                          ("mlp.layers.0", (hid, do)), ("mlp.layers.1", (do, hid))):
            sd[p + n + ".weight"] = tn((o, i))
            sd[p + n + ".bias"] = 0.02 * torch.randn(o, generator=g, dtype=dtype)
        if d != do:
            sd[p + "proj.weight"] = tn((do, d))
            sd[p + "proj.bias"] = 0.02 * torch.randn(do, generator=g, dtype=dtype)
    chans = [D * 2, D * 4, D * 8]
    tot = sum(chans)
    sd["fusion.conv1x1.weight"] = uni((512, tot, 1, 1), tot)
    bn("fusion.bn.", 512)
    sd["fusion.se_block.fc.0.weight"] = uni((32, 512), 512)
    sd["fusion.se_block.fc.2.weight"] = uni((512, 32), 32)
    sd["context.reduce.0.weight"] = uni((128, 512, 1, 1), 512)
    bn("context.reduce.1.", 128)
    for i in range(4):
        sd[f"context.branches.{i}.0.weight"] = uni((128, 1, 3, 3), 9)
        bn(f"context.branches.{i}.1.", 128)
    sd["context.global_branch.1.weight"] = uni((128, 128, 1, 1), 128)
    bn("context.global_branch.2.", 128)
    sd["context.fusion.0.weight"] = uni((128, 5, 1, 1), 5)
    bn("context.fusion.1.", 128)
    sd["context.expand.0.weight"] = uni((256, 128, 1, 1), 128)
    bn("context.expand.1.", 256)
    sd["edge_detector.conv1.weight"] = uni((64, 256, 3, 3), 256 * 9)
    bn("edge_detector.bn1.", 64)
    sd["edge_detector.edge_conv.weight"] = uni((1, 64, 1, 1), 64)
    sd["edge_detector.edge_conv.bias"] = uni((1,), 64)
    prev = 256
    for i, (c, ec) in enumerate(zip((256, 128, 64), (64, 64, 0))):
        p = f"decoder.decoder_blocks.{i}."
        sd[p + "conv1.weight"] = uni((c, prev + ec, 3, 3), (prev + ec) * 9)
        sd[p + "conv1.bias"] = uni((c,), (prev + ec) * 9)
        bn(p + "bn1.", c)
        sd[p + "conv2.weight"] = uni((c, c, 3, 3), c * 9)
        sd[p + "conv2.bias"] = uni((c,), c * 9)
        bn(p + "bn2.", c)
        sd[f"decoder.pred_heads.{i}.weight"] = uni((1, c, 1, 1), c)
        sd[f"decoder.pred_heads.{i}.bias"] = uni((1,), c)
        prev = c
    return sd


def is_buffer_key(k: str) -> bool:
    return k.endswith(("running_mean", "running_var", "num_batches_tracked"))


def synthetic_batch(B: int, S: int, seed: int = 0):
    """SURVEY §8(d) synthetic inputs: images randn seed s, masks rand>0.7 seed s+1, edges rand>0.95 seed s+2."""
    g0, g1, g2 = (torch.Generator().manual_seed(seed + i) for i in range(3))
    images = torch.randn(B, 3, S, S, generator=g0)
    masks = [(torch.rand(1, S, S, generator=g1) > 0.7).float() for _ in range(B)]
    edges = [(torch.rand(1, S, S, generator=g2) > 0.95).float() for _ in range(B)]
    return images, masks, edges


# --------------------------------------------------------------------------------------------
# One training step (engine/trainer.py:255-306 param groups, :308-427 step)
# --------------------------------------------------------------------------------------------
def param_groups(names: Sequence[str], base_lr=1e-4, wd=1e-5, enc_ratio=0.05):
    """name -> (lr, weight_decay) following Trainer._get_param_groups, including the quirk that BN
    layers living in nn.Sequential (no 'bn'/'norm' in their name) DO get weight decay."""
    out = {}
    for n in names:
        if "encoder" in n:
            out[n] = (base_lr * enc_ratio, 0.0)
        elif "norm" in n or "bn" in n:
            out[n] = (base_lr, 0.0)
        else:
            out[n] = (base_lr, wd)
    return out


def train_step(sd: SD, opt_state: dict, images, masks, edges, loss_cfg=LOSS_DEFAULT_YAML, base_lr=1e-4, wd=1e-5,
               enc_ratio=0.05, clip=1.0, cfg=HIERA_L, betas=(0.9, 0.999), eps=1e-8):
    """fp32 forward + CODLoss + backward + global-norm clip + AdamW (no AMP: bf16/fp16 autocast is a
    precision policy, not part of the algorithm).  Updates sd / opt_state in place; returns losses."""
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not is_buffer_key(k)}
    out = spegnet_forward(sd, images, training=True, cfg=cfg)
    losses = cod_loss(out["predictions"], out["edge"], masks, edges, **loss_cfg)
    grads = torch.autograd.grad(losses["loss"], list(params.values()), allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params.values())]
    with torch.no_grad():
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
        coef = torch.clamp(clip / (total + 1e-6), max=1.0) if clip > 0 else 1.0
        step = opt_state.setdefault("step", 0) + 1
        opt_state["step"] = step
        groups = param_groups(list(params.keys()), base_lr, wd, enc_ratio)
        for (k, p), g in zip(params.items(), grads):
            lr, w = groups[k]
            g = g * coef
            m = opt_state.setdefault("m." + k, torch.zeros_like(p))
            v = opt_state.setdefault("v." + k, torch.zeros_like(p))
            p.mul_(1 - lr * w)
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
    for p in params.values():
        p.requires_grad_(False)
    return {k: float(v) for k, v in losses.items()}, float(total), grads


def preprocess_image(img_u8_hwc: torch.Tensor, size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """CODImageProcessor.process_image after decoding (reference utils/image_processor.py:118-131): uint8 HWC -> float / 255 ->
    F.interpolate(bilinear, align_corners=False, antialias=True) -> (v - mean) / std.  CPU, fp32."""
    t = img_u8_hwc.cpu().float().permute(2, 0, 1) / 255.0
    size = (size, size) if isinstance(size, int) else tuple(size)
    t = F.interpolate(t[None], size=size, mode="bilinear", align_corners=False, antialias=True)[0]
    return (t - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
