"""Host-side image pre-processing for Predictor (reference utils/image_processor.py:48-212): PIL RGB -> float/255 ->
antialiased bilinear resize to SxS -> ImageNet normalisation; masks / edges -> {0,1} at their original size.
This is I/O-side host code (outside the kernel scope), kept so the predictor entry point works;
process_image_device is the fused HIP version of the same arithmetic."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


@dataclass
class ProcessedSample:
    image: torch.Tensor
    mask: Optional[torch.Tensor] = None
    edge: Optional[torch.Tensor] = None


class CODImageProcessor:
    def __init__(self, target_size: int = 512, normalize_mean: Tuple[float, float, float] = (0.485, 0.456, 0.406),
                 normalize_std: Tuple[float, float, float] = (0.229, 0.224, 0.225)):
        self.target_size = (target_size, target_size)
        self.norm_mean = torch.tensor(normalize_mean).view(-1, 1, 1)
        self.norm_std = torch.tensor(normalize_std).view(-1, 1, 1)

    @torch.no_grad()
    def process_image(self, image_path: Union[str, Path]) -> torch.Tensor:
        try:
            img = Image.open(str(image_path)).convert('RGB')
        except Exception as e:  # same error contract as the reference
            raise RuntimeError(f"Failed to process image {image_path}: {e}")
        t = torch.from_numpy(np.array(img)).float().permute(2, 0, 1) / 255.0
        t = F.interpolate(t[None], size=self.target_size, mode='bilinear', align_corners=False, antialias=True)[0]
        return (t - self.norm_mean) / self.norm_std

    @torch.no_grad()
    def process_image_device(self, image_path: Union[str, Path], device="cuda") -> torch.Tensor:
        """Same result as process_image, computed on the GPU: the decoded uint8 HWC image is uploaded as is (3 bytes per pixel) and
        one HIP kernel does /255, the antialiased resize and the normalisation (ops.preprocess_image)."""
        from .. import ops
        try:
            img = Image.open(str(image_path)).convert('RGB')
        except Exception as e:
            raise RuntimeError(f"Failed to process image {image_path}: {e}")
        u8 = torch.from_numpy(np.array(img)).to(device, non_blocking=True)
        return ops.preprocess_image(u8, self.target_size, self.norm_mean.flatten().tolist(), self.norm_std.flatten().tolist())

    @torch.no_grad()
    def process_mask(self, path: Union[str, Path]) -> torch.Tensor:
        m = torch.from_numpy(np.array(Image.open(str(path)).convert('L'))).float()
        return (m > 127.5).float()[None]

    def __call__(self, image_path, mask_path=None, edge_path=None) -> ProcessedSample:
        return ProcessedSample(self.process_image(image_path), self.process_mask(mask_path) if mask_path else None,
                               self.process_mask(edge_path) if edge_path else None)
