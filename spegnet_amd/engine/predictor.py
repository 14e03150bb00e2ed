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
from .distributed import graph_capture_mode
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

    def _forward_batch(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One eval-mode forward of [n,3,S,S].  Full batches (n == batch_size) replay a hipGraph captured on first use (BASELINE config #3:
        no per-kernel launch cost); a shorter last batch runs eagerly."""
        n = x.shape[0]
        if n != self.batch_size or self.batch_size < 2:
            with torch.no_grad():
                return self.model(x)
        if getattr(self, "_graph", None) is None or self._static_x.shape != x.shape:
            with torch.no_grad():
                self._static_x = x.clone()
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self.model(self._static_x)                 # warm-up off the capture stream (allocator, lazy init)
                torch.cuda.current_stream(self.device).wait_stream(side)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, capture_error_mode=graph_capture_mode()):
                    self._static_out = self.model(self._static_x)
        self._static_x.copy_(x)
        self._graph.replay()
        return self._static_out

    def predict_batch(self, image_paths: List[str], output_size: Optional[Tuple[int, int]] = None) -> Dict:
        """Reference engine/predictor.py:376-430 walks the list in chunks of batch_size and still predicts image by image; here a chunk
        IS a batch: the decoded uint8 images are uploaded together, resized / normalised by one batched HIP launch
        (ops.preprocess_batch), and go through ONE forward (a captured hipGraph for full batches).  Per-image results are those of
        predict_single (eval-mode BatchNorm: no cross-sample coupling)."""
        from .. import ops
        self.result_manager.log_message(f"Starting batch prediction of {len(image_paths)} images with batch size {self.batch_size}")
        ip = self.image_processor
        mean, std = ip.norm_mean.flatten().tolist(), ip.norm_std.flatten().tolist()
        bs = max(int(self.batch_size or 1), 1)
        for i in range(0, len(image_paths), bs):
            paths = image_paths[i:i + bs]
            t0 = time.time()
            originals = []
            for p in paths:
                try:
                    originals.append(np.array(Image.open(p).convert('RGB')))
                except Exception as e:
                    raise RuntimeError(f"Failed to process image {p}: {e}")
            sizes = [(o.shape[0], o.shape[1]) for o in originals]
            offs, off = [], 0
            for h, w in sizes:
                offs.append(off)
                off += (h * w * 3 + 255) // 256 * 256
            host = torch.empty(off, dtype=torch.uint8).pin_memory()
            for o, k in zip(originals, offs):
                host[k:k + o.size].copy_(torch.from_numpy(o).reshape(-1))
            x = ops.preprocess_batch(host.to(self.device, non_blocking=True), offs, sizes, ip.target_size, mean, std)
            self.result_manager.update_timing('preprocessing', time.time() - t0)
            t0 = time.time()
            out = self._forward_batch(x)
            torch.cuda.synchronize(self.device)
            self.result_manager.update_timing('inference', time.time() - t0)
            t0 = time.time()
            seg, edge = out['predictions'][-1].float(), out['edge'].float()
            if output_size:
                seg = F.interpolate(seg, size=output_size, mode='bilinear', align_corners=False)
                edge = F.interpolate(edge, size=output_size, mode='bilinear', align_corners=False)
            seg_np, edge_np = seg.sigmoid().squeeze(1).cpu().numpy(), edge.sigmoid().squeeze(1).cpu().numpy()
            self.result_manager.update_timing('postprocessing', time.time() - t0)
            for j, p in enumerate(paths):
                self.result_manager.save_prediction(Path(p).name, seg_np[j], edge_np[j], originals[j])
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
