"""SPEGNet on MI355X: the reference's nn.Module surface over the hand-written HIP path.

Mirrors models/spegnet.py:47-206 of the reference: `SPEGNet(config)` with `config['encoder']` keys
`config_path | checkpoint_path | variant`, attributes `encoder / fusion / context / edge_detector / decoder /
in_channels_list`, `forward(x[B,3,H,W]) -> {'predictions': [p1,p2,p3], 'edge', 'features': {'context','fused',
'edge_features'}}`, `ValueError` unless H,W % 32 == 0 (models/feature_encoding.py:230-233), and the same state_dict
keys.  The sub-modules are parameter containers; the arithmetic is `Engine` (HIP kernels through the C ABI).

Extra keys understood in `config` (all optional): `compute_dtype` ('bf16' default | 'fp32' parity mode).
There is no CPU implementation: a non-CUDA input raises.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .engine import Engine
from .params import BUFFER_KINDS, HIERA_CONFIGS, ParamTree, build_tree, param_specs

logger = logging.getLogger(__name__)
_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}


class HieraSAM2FeatureEncoder(ParamTree):
    """Parameter container + metadata of the Hiera trunk (reference: models/feature_encoding.py:111-290)."""
    channels_dict = {
        'tiny': [96, 192, 384, 768], 'small': [96, 192, 384, 768], 'base': [96, 192, 384, 768],
        'base_plus': [112, 224, 448, 896], 'large': [144, 288, 576, 1152], 'huge': [256, 512, 1024, 2048],
        'test_tiny': [16, 32, 64, 128],
    }

    def __init__(self, variant: str = 'large'):
        super().__init__()
        if variant not in self.channels_dict:
            raise ValueError(f"Invalid variant. Choose from: {list(self.channels_dict.keys())}")
        if variant not in HIERA_CONFIGS:
            raise ValueError(f"variant '{variant}' has no block layout in this build ({list(HIERA_CONFIGS)})")
        self.variant = variant
        self._owner = None

    @property
    def channels(self) -> List[int]:
        return self.channels_dict[self.variant]

    @property
    def param_count(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def get_output_shapes(self, height: int, width: int) -> List[Tuple[int, int, int]]:
        if height % 32 != 0 or width % 32 != 0:
            raise ValueError("Input dimensions must be divisible by 32")
        return [(c, height // (4 * 2 ** i), width // (4 * 2 ** i)) for i, c in enumerate(self.channels)]

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """[B,3,H,W] -> 4 stage maps, NCHW-shaped (channels-last memory)."""
        if self._owner is None:
            raise RuntimeError("encoder is not attached to a SPEGNet")
        return self._owner().encode(x)

    def __repr__(self) -> str:
        return f"HieraSAM2Encoder(\n  variant={self.variant}\n  channels={self.channels}\n  params={self.param_count:,}\n)"


def _check_input(x: torch.Tensor):
    if x.dim() != 4:
        raise ValueError(f"Expected 4D input (B,C,H,W), got {x.dim()}D")
    if any(s % 32 != 0 for s in x.shape[-2:]):
        raise ValueError("Input spatial dims must be divisible by 32")
    if not x.is_cuda:
        raise RuntimeError("spegnet_amd.SPEGNet runs only on an MI355X (HIP) device; there is no CPU fallback")


class _SpegnetFn(torch.autograd.Function):
    """One autograd node for the whole network.  Parameter gradients are written straight into the (flat) .grad
    buffers by the backward kernels; autograd only carries d(outputs) in and d(image) (None) out."""

    @staticmethod
    def forward(ctx, model, x, hook):
        eng = model.engine
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward (no host-side zero checks)
        with ops.cu_budget(model.cu_budget):
            feats, tctx = eng.trunk_fwd(x, True, True)
            out, hctx = eng.head_fwd(feats[1:4], True, True)
        ctx.model, ctx.tctx, ctx.hctx = model, tctx, hctx
        ctx.feat_shapes = [f.shape for f in feats]
        p = out["predictions"]
        return (p[0], p[1], p[2], out["edge"], out["context"], out["fused"], out["edge_features"], hook.new_zeros(()))

    @staticmethod
    def backward(ctx, g1, g2, g3, ge, gc, gf, gef, _gh):
        model = ctx.model
        eng = model.engine
        extra = {k: v for k, v in (("context", gc), ("fused", gf), ("edge_features", gef)) if v is not None} or None
        with ops.cu_budget(model.cu_budget):   # (runs on autograd's thread: the budget travels with the model, not with a thread)
            d = eng.head_bwd(ctx.hctx, [g1, g2, g3], ge, extra)
            eng.trunk_bwd(ctx.tctx, [None] + d)
        ctx.tctx = ctx.hctx = None
        if model._post_backward is not None:
            model._post_backward()
        return None, None, None


class SPEGNet(nn.Module):
    def __init__(self, config: Dict):
        super().__init__()
        enc_cfg = config['encoder']
        variant = enc_cfg.get('variant', 'large')
        self.encoder = HieraSAM2FeatureEncoder(variant)
        self.hiera_cfg = HIERA_CONFIGS[variant]
        self.compute_dtype = _DTYPES[str(config.get('compute_dtype', 'bf16')).lower()]
        encoder_channels = self.encoder.channels
        self.in_channels_list = encoder_channels[1:4]
        # parameters / buffers under the reference's names: encoder.encoder.*, fusion.*, context.*,
        # edge_detector.*, decoder.*
        kinds = {}
        g = torch.Generator().manual_seed(int(config.get('init_seed', 0)))
        empty = str(config.get('init', 'random')) == 'empty'     # parameters left uninitialised: a load_state_dict follows
        from .params import init_tensor
        for name, shape, kind in param_specs(self.hiera_cfg):
            top, _, rest = name.partition(".")
            if top not in self._modules:
                self.add_module(top, ParamTree())
            self._modules[top].insert(rest, init_tensor(name, shape, kind, g, empty), kind in BUFFER_KINDS)
            kinds[name] = kind
        self._kinds = kinds
        import weakref
        self.encoder._owner = weakref.ref(self)
        self._engine: Optional[Engine] = None
        self._packed_version = -1
        self._param_version = 0
        self._post_backward = None
        self._hook = None
        self.cu_budget = 0     # CUs the GEMM grids of this model's eager forward / backward are sized for (0 = all)
        ckpt = enc_cfg.get('checkpoint_path')
        if ckpt and os.path.exists(ckpt):
            self.load_encoder_checkpoint(ckpt)
        elif ckpt:
            logger.warning("SAM2 checkpoint %s not found: trunk keeps its random initialisation", ckpt)

    # ---------------------------------------------------------------------------------------------
    def load_encoder_checkpoint(self, path: str):
        """Loads the Hiera trunk weights from a SAM2 checkpoint (keys image_encoder.trunk.*)."""
        sd = torch.load(path, map_location="cpu", weights_only=False)
        sd = sd.get("model", sd)
        pre = "image_encoder.trunk."
        mine = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        missing, unexpected = self.encoder.load_state_dict({"encoder." + k: v for k, v in mine.items()}, strict=False)
        if missing:
            logger.warning("trunk keys missing from %s: %s", path, missing[:5])
        self.mark_params_changed()

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self, self.hiera_cfg, self.compute_dtype)
        if self._packed_version != self._param_version:
            self._engine.refresh_params()
            self._engine.pack()
            self._packed_version = self._param_version
        return self._engine

    def mark_params_changed(self):
        """Call after the fp32 parameters changed (optimizer step, load_state_dict): the compute-dtype weight
        copies are re-packed lazily before the next forward."""
        self._param_version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.mark_params_changed()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engine = None
        self.mark_params_changed()
        return r

    # ---------------------------------------------------------------------------------------------
    def encode(self, x: torch.Tensor) -> List[torch.Tensor]:
        _check_input(x)
        with torch.no_grad():
            feats, _ = self.engine.trunk_fwd(x, False, False)
        return [f.permute(0, 3, 1, 2) for f in feats]

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        _check_input(x)
        if self.training and torch.is_grad_enabled():
            if self._hook is None or self._hook.device != x.device:
                self._hook = torch.zeros((), device=x.device, requires_grad=True)
            p1, p2, p3, edge, context, fused, edge_f, _ = _SpegnetFn.apply(self, x, self._hook)
        else:
            with torch.no_grad():
                eng = self.engine
                feats, _ = eng.trunk_fwd(x, self.training, False)
                out, _ = eng.head_fwd(feats[1:4], self.training, False)
            p1, p2, p3 = out["predictions"]
            edge, context, fused, edge_f = out["edge"], out["context"], out["fused"], out["edge_features"]
        return {
            'predictions': [p1, p2, p3],
            'edge': edge,
            'features': {'context': context.permute(0, 3, 1, 2), 'fused': fused.permute(0, 3, 1, 2),
                         'edge_features': edge_f.permute(0, 3, 1, 2)},
        }
