"""Inference entry point for the MI355X path, mirroring reference engine/predictor.py:217-467:
`Predictor(model_path, model_config, dir_manager, output_dir=None, device=None, batch_size=1)`,
`.predict_single(path, output_size) -> (seg_np, edge_np, original_image)`, `.predict_batch`, `.predict_directory`.

The forward runs on the HIP kernels (bf16 by default, `compute_dtype: fp32` for parity mode); resize of the two
logit maps to `output_size` and the sigmoid are one-image post-processing left to torch.  Writing the six
visualisations needs cv2 (absent here) and is delegated to an optional `result_manager`.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from ..models.spegnet import SPEGNet
from ..utils.image_processor import CODImageProcessor


class _Timing:
    def __init__(self):
        self.t: Dict[str, List[float]] = {}
        self.messages: List[str] = []

    def update_timing(self, k, v):
        self.t.setdefault(k, []).append(v)

    def log_message(self, m):
        self.messages.append(m)
        logging.info(m)

    def save_prediction(self, *a, **k):
        pass

    def summarize(self):
        return {"total_predictions": len(self.t.get("inference", [])),
                "average_timings": {k: float(np.mean(v)) for k, v in self.t.items()},
                "total_processing_time": float(sum(sum(v) for v in self.t.values()))}


class Predictor:
    def __init__(self, model_path: str, model_config: Dict, dir_manager=None, output_dir: Optional[str] = None,
                 device: Optional[str] = None, batch_size: Optional[int] = 1, result_manager=None):
        if not Path(model_path).exists():
            raise FileNotFoundError(f"Model checkpoint not found: {model_path}")
        self.device = torch.device(device or ('cuda' if torch.cuda.is_available() else 'cpu'))
        if self.device.type != 'cuda':
            raise RuntimeError("spegnet_amd.Predictor needs an MI355X (HIP) device; there is no CPU fallback")
        self.batch_size = batch_size
        img_config = model_config['image_processing']
        self.image_processor = CODImageProcessor(target_size=img_config['target_size'],
                                                 normalize_mean=tuple(img_config['normalize_mean']),
                                                 normalize_std=tuple(img_config['normalize_std']))
        self.result_manager = result_manager or _Timing()
        self.model = self._initialize_model(model_path, model_config)
        self.result_manager.log_message(f"Model loaded from: {model_path}")

    def _initialize_model(self, model_path: str, model_config: Dict) -> SPEGNet:
        model = SPEGNet(model_config)
        checkpoint = torch.load(model_path, map_location='cpu', weights_only=False)
        model.load_state_dict(checkpoint['model_state_dict'])
        model = model.to(self.device).eval()
        s = model_config['image_processing']['target_size']
        with torch.inference_mode():                      # warm-up, as the reference does (predictor.py:283-288)
            model(torch.randn(self.batch_size, 3, s, s, device=self.device))
        torch.cuda.synchronize(self.device)
        return model

    def preprocess_image(self, image_path: str) -> torch.Tensor:
        t0 = time.time()
        if torch.device(self.device).type == "cuda":   # decode on the host, everything else in one HIP kernel (ops.preprocess_image)
            t = self.image_processor.process_image_device(image_path, self.device)
        else:
            t = self.image_processor(image_path).image.to(self.device)
        self.result_manager.update_timing('preprocessing', time.time() - t0)
        return t.unsqueeze(0)

    def predict_single(self, image_path: str, output_size: Optional[Tuple[int, int]] = None):
        x = self.preprocess_image(image_path)
        with torch.no_grad():
            t0 = time.time()
            out = self.model(x)
            torch.cuda.synchronize(self.device)
            dt = time.time() - t0
            self.result_manager.update_timing('inference', dt)
            self.result_manager.log_message(f"Inference time for {image_path}: {dt:.3f}s")
            seg, edge = out['predictions'][-1].float(), out['edge'].float()
            t1 = time.time()
            if output_size:
                seg = F.interpolate(seg, size=output_size, mode='bilinear', align_corners=False)
                edge = F.interpolate(edge, size=output_size, mode='bilinear', align_corners=False)
            seg_np = seg.sigmoid().squeeze().cpu().numpy()
            edge_np = edge.sigmoid().squeeze().cpu().numpy()
            self.result_manager.update_timing('postprocessing', time.time() - t1)
        original = np.array(Image.open(image_path).convert('RGB'))
        return seg_np, edge_np, original

    def predict_batch(self, image_paths: List[str], output_size: Optional[Tuple[int, int]] = None) -> Dict:
        self.result_manager.log_message(f"Starting batch prediction of {len(image_paths)} images with batch size {self.batch_size}")
        for p in image_paths:
            seg, edge, orig = self.predict_single(p, output_size)
            self.result_manager.save_prediction(Path(p).name, seg, edge, orig)
        return self.result_manager.summarize()

    def predict_directory(self, input_dir: str, output_size: Optional[Tuple[int, int]] = None,
                          extensions: tuple = ('.jpg', '.png', '.jpeg')) -> Dict:
        d = Path(input_dir)
        if not d.is_dir():
            raise NotADirectoryError(f"Invalid directory: {input_dir}")
        paths = [str(p) for p in d.glob('**/*') if p.suffix.lower() in extensions]
        if not paths:
            raise ValueError(f"No valid images found in {input_dir}")
        self.result_manager.log_message(f"Found {len(paths)} images in {input_dir}")
        return self.predict_batch(paths, output_size)
