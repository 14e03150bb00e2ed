"""Evaluation entry point for the MI355X path, with the surface of the reference's engine/evaluator.py:266-663:
`Evaluator(model_path, dir_manager, model_config, device, batch_size)`, `.evaluate(test_loader, dataset_name) -> {'s_alpha',
'weighted_f','mae','e_phi','mean_f'}` (averages over the dataset), per-sample good / medium / bad bucketing (:133-152) and an
`evaluation_summary.json` with metrics, timing and the category distribution (:597-635).

MI355X-first differences: the forward runs on the HIP kernels (a hipGraph replay for full batches, as Predictor.predict_batch), and the
five COD measures are evaluated ON THE DEVICE (utils/metrics.py) instead of copying every map to a pool of CPU workers running
`py_sod_metrics` (reference utils/metrics.py:142-260).  Writing the per-sample PNG visualisations needs cv2 (absent, SURVEY.md 2 row
12) and is delegated to an optional `result_manager.save_prediction`.

Reference quirk kept on purpose (`double_sigmoid=True`): Evaluator._process_batch passes sigmoid(resized logits) to
compute_metrics (:539-560), which applies sigmoid AGAIN before quantising (utils/metrics.py:205) -- the published numbers are metrics
of sigmoid(sigmoid(z)).  `double_sigmoid=False` evaluates sigmoid(z), which is what Trainer.validate's metrics see."""
from __future__ import annotations

import json
import logging
import os
import time
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..models.spegnet import SPEGNet
from .distributed import graph_capture_mode
from ..utils.metrics import MetricsProcessor

METRIC_KEYS = ('s_alpha', 'weighted_f', 'mae', 'e_phi', 'mean_f')


def determine_quality_category(metrics: Dict[str, float]) -> str:
    """good / medium / bad: both the structure measure and the weighted F-measure must reach 0.8 / 0.6 (reference :133-152)"""
    s, f = metrics['s_alpha'], metrics['weighted_f']
    if s >= 0.8 and f >= 0.8:
        return 'good'
    if s >= 0.6 and f >= 0.6:
        return 'medium'
    return 'bad'


class Evaluator:
    def __init__(self, model_path: str, dir_manager, model_config: Dict, device: torch.device, batch_size: int, result_manager=None,
                 double_sigmoid: bool = True):
        if not Path(model_path).exists():
            raise FileNotFoundError(f"Model checkpoint not found: {model_path}")
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("spegnet_amd.Evaluator needs an MI355X (HIP) device; there is no CPU fallback")
        self.dir_manager = dir_manager
        self.batch_size = int(batch_size)
        self.double_sigmoid = double_sigmoid
        self.result_manager = result_manager
        self.metrics_processor = MetricsProcessor()
        self.timing_stats = {'inference_times': [], 'processing_times': [], 'total_time': 0.0, 'total_samples': 0}
        self.categories: Dict[str, Dict[str, int]] = {}
        self.model = self._load_model(model_path, model_config, self.batch_size)
        self._graph = None

    def _load_model(self, model_path: str, model_config: Dict, batch_size: int) -> SPEGNet:
        model = SPEGNet(model_config)
        ckpt = torch.load(model_path, map_location='cpu', weights_only=False)
        model.load_state_dict(ckpt['model_state_dict'])
        model = model.to(self.device).eval()
        s = model_config['image_processing']['target_size']
        with torch.inference_mode():                       # 3 warm-ups, as the reference (:357-363)
            for _ in range(3):
                model(torch.randn(batch_size, 3, s, s, device=self.device))
        torch.cuda.synchronize(self.device)
        return model

    def _forward(self, images: torch.Tensor) -> Dict[str, torch.Tensor]:
        """full batches replay one captured hipGraph; anything else runs eagerly"""
        if images.shape[0] != self.batch_size or self.batch_size < 2:
            with torch.no_grad():
                return self.model(images)
        if self._graph is None or self._static_x.shape != images.shape:
            with torch.no_grad():
                self._static_x = images.clone()
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self.model(self._static_x)
                torch.cuda.current_stream(self.device).wait_stream(side)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, capture_error_mode=graph_capture_mode()):
                    self._static_out = self.model(self._static_x)
        self._static_x.copy_(images)
        self._graph.replay()
        return self._static_out

    def _process_batch(self, dataset_name: str, batch: Dict) -> Tuple[Dict[str, float], int]:
        t0 = time.time()
        images = batch['images'].to(self.device, non_blocking=True)
        masks = [m.to(self.device, non_blocking=True) for m in batch['masks']]
        names = batch.get('names') or [str(i) for i in range(len(masks))]
        ti = time.time()
        out = self._forward(images)
        torch.cuda.synchronize(self.device)
        self.timing_stats['inference_times'].append(time.time() - ti)
        seg, edge = out['predictions'][-1].float(), out['edge'].float()
        tot = {k: 0.0 for k in METRIC_KEYS}
        cats = self.categories.setdefault(dataset_name, {'good': 0, 'medium': 0, 'bad': 0})
        for i, name in enumerate(names):
            z = F.interpolate(seg[i:i + 1], size=masks[i].shape[-2:], mode='bilinear', align_corners=False)
            arg = z.sigmoid() if self.double_sigmoid else z     # compute_metrics applies sigmoid itself (reference quirk: see module doc)
            m = self.metrics_processor.compute_metrics(seg_pred=arg, seg_gt=[masks[i]])
            for k in METRIC_KEYS:
                tot[k] += m[k]
            cats[determine_quality_category(m)] += 1
            if self.result_manager is not None:
                e = F.interpolate(edge[i:i + 1], size=masks[i].shape[-2:], mode='bilinear', align_corners=False).sigmoid()
                self.result_manager.save_prediction(name, z.sigmoid(), e, m, dataset_name)
        self.timing_stats['processing_times'].append(time.time() - t0)
        return tot, len(names)

    def evaluate(self, test_loader, dataset_name: str) -> Dict[str, float]:
        total = {k: 0.0 for k in METRIC_KEYS}
        n = 0
        t0 = time.time()
        logging.info("Starting evaluation of %s ...", dataset_name)
        for batch in test_loader:
            bm, bs = self._process_batch(dataset_name, batch)
            for k in METRIC_KEYS:
                total[k] += bm[k]
            n += bs
        if n == 0:
            raise ValueError(f"no samples in the {dataset_name} loader")
        avg = {k: v / n for k, v in total.items()}
        self.timing_stats['total_time'] = time.time() - t0
        self.timing_stats['total_samples'] = n
        self._save_evaluation_summary(dataset_name, avg)
        return avg

    def _save_evaluation_summary(self, dataset_name: str, metrics: Dict[str, float]) -> Dict:
        ts = self.timing_stats
        summary = {'metrics': metrics,
                   'timing': {'total_time': ts['total_time'], 'avg_inference_time': float(np.mean(ts['inference_times'])),
                              'avg_processing_time': float(np.mean(ts['processing_times'])), 'total_samples': ts['total_samples']},
                   'categories': self.categories.get(dataset_name, {})}
        root = getattr(self.dir_manager, "run_dir", None) if self.dir_manager is not None else None
        if root is not None:
            d = os.path.join(str(root), dataset_name)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'evaluation_summary.json'), 'w') as f:
                json.dump(summary, f, indent=4)
        logging.info("%s: %s", dataset_name, {k: round(v, 4) for k, v in metrics.items()})
        return summary
