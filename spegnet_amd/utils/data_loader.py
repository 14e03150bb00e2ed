"""Dataset + loaders for the MI355X path, with the surface of the reference's utils/data_loader.py
(CODDataset :65-175, collate_fn :177-212, get_training_loaders :214-314, get_test_loaders :316-375):
`root/{Imgs,GT[,Edges]}` layout, images resized + normalised to the model size, masks / edges kept at their
original size as lists, 90/10 split with seed 42, `{'images','masks','edges'|'names'}` batches.

What differs by design (the device input pipeline):
  * `device_preprocess=True`: a sample carries the DECODED uint8 HWC image (3 bytes per pixel instead of 12 bytes of float CHW per
    resized pixel); `collate_fn` keeps them as a list and `DeviceBatcher` uploads them through pinned staging buffers on a copy
    stream and runs ONE batched HIP launch (ops.preprocess_batch: /255, ATen-exact antialiased bilinear resize, normalise) that writes
    the [B,3,S,S] model input -- the reference does that arithmetic per image in DataLoader worker processes;
  * `pin_memory=True` loaders, and `prefetch()` (engine/trainer.py) overlaps the next batch's H2D with the current step.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from PIL import Image
from torch.utils.data import ConcatDataset, DataLoader, Dataset

from .image_processor import CODImageProcessor


class CODDataset(Dataset):
    """One dataset split directory: Imgs/*.jpg|png, GT/*.png and (training) Edges/*.png, matched by file stem."""

    def __init__(self, root_dir: str, processor_params: Dict, is_train: bool = True, device_preprocess: bool = False):
        self.is_train = is_train
        self.device_preprocess = device_preprocess
        self.image_dir = os.path.join(root_dir, 'Imgs')
        self.mask_dir = os.path.join(root_dir, 'GT')
        self.edge_dir = os.path.join(root_dir, 'Edges') if is_train else None
        if not os.path.exists(self.image_dir) or not os.path.exists(self.mask_dir):
            raise FileNotFoundError(f"Required directories not found in {root_dir}")
        if is_train and not os.path.exists(self.edge_dir):
            raise FileNotFoundError(f"Edge directory not found for training in {root_dir}")
        self.processor = CODImageProcessor(**processor_params)
        self.filenames = self._valid_files()

    def _valid_files(self) -> List[str]:
        stems = lambda d, ext: {f.split('.')[0] for f in os.listdir(d) if f.endswith(ext)}
        ok = stems(self.image_dir, ('.jpg', '.png')) & stems(self.mask_dir, '.png')
        if self.is_train:
            ok &= stems(self.edge_dir, '.png')
        ok = sorted(ok)
        if not ok:
            raise ValueError(f"No valid samples found in {self.image_dir}")
        return ok

    def __len__(self) -> int:
        return len(self.filenames)

    def __getitem__(self, idx: int) -> Dict:
        name = self.filenames[idx]
        img_path = os.path.join(self.image_dir, name + ".jpg")
        if not os.path.exists(img_path):
            img_path = os.path.join(self.image_dir, name + ".png")
        mask_path = os.path.join(self.mask_dir, name + ".png")
        if self.device_preprocess:
            try:
                img = torch.from_numpy(np.array(Image.open(img_path).convert('RGB')))     # uint8 [H, W, 3]: resized on the device
            except Exception as e:
                raise RuntimeError(f"Failed to process image {img_path}: {e}")
        else:
            img = self.processor.process_image(img_path)
        sample = {'image': img, 'mask': self.processor.process_mask(mask_path)}
        if self.is_train:
            sample['edge'] = self.processor.process_mask(os.path.join(self.edge_dir, name + ".png"))
        else:
            sample['name'] = name
        return sample


def collate_fn(batch: List[Dict]) -> Dict:
    """images stacked [B,3,S,S] (or a list of uint8 HWC tensors in device_preprocess mode); masks / edges stay lists."""
    if not batch:
        raise ValueError("Empty batch received")
    first = batch[0]['image']
    if first.dtype == torch.uint8:
        out = {'images_u8': [b['image'] for b in batch]}
    else:
        out = {'images': torch.stack([b['image'] for b in batch])}
    out['masks'] = [b['mask'] for b in batch]
    if 'edge' in batch[0]:
        out['edges'] = [b['edge'] for b in batch]
    else:
        out['names'] = [b['name'] for b in batch]
    return out


def _processor_params(model_config: Dict) -> Dict:
    ip = model_config['image_processing']
    return {'target_size': ip['target_size'], 'normalize_mean': tuple(ip['normalize_mean']), 'normalize_std': tuple(ip['normalize_std'])}


class _StridedShard(torch.utils.data.Sampler):
    """Unshuffled rank shard WITHOUT padding: indices rank, rank + world, ...  (DistributedSampler pads a shard with repeated samples so
    that all ranks get the same count; the all-reduced validation sums would then count up to world - 1 samples twice.)  Shards may
    differ in length by one: the validation loop has no collective per batch, only the all-reduce of the totals at its end."""

    def __init__(self, n: int, rank: int, world: int):
        self.idx = list(range(rank, n, world))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)

    def set_epoch(self, epoch: int) -> None:   # (same interface as DistributedSampler; nothing to reshuffle)
        pass


def _loader(ds, batch_size, shuffle, num_workers, rank: int = 0, world: int = 1, drop_last: bool = False) -> DataLoader:
    """world > 1: each rank iterates its own shard (the reference's loader, utils/data_loader.py:287-301, is single-GPU; this is SURVEY
    8(e)'s addition to it): training = DistributedSampler (seed 42; the caller calls `loader.sampler.set_epoch(epoch)` once per epoch),
    validation / test (shuffle False) = an unpadded strided shard, so that job-wide sums count every sample exactly once."""
    sampler = None
    if world > 1:
        if shuffle:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=42, drop_last=drop_last)
        else:
            sampler = _StridedShard(len(ds), rank, world)
        shuffle = False
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=num_workers, collate_fn=collate_fn,
                      pin_memory=torch.cuda.is_available(), persistent_workers=num_workers > 0, drop_last=drop_last)


def get_training_loaders(dataset_dirs: List[str], model_config: Dict, batch_size: int = 16, num_workers: int = 4, val_ratio: float = 0.1,
                         device: str = 'cuda', device_preprocess: bool = False, rank: int = 0, world: int = 1,
                         drop_last: bool = False) -> Tuple[DataLoader, Optional[DataLoader]]:
    """rank / world: this process's shard of both splits (the 90/10 split itself is seeded and identical on every rank).  drop_last
    applies to the TRAINING loader only (fixed-shape captured steps, equal step counts across ranks)."""
    params = _processor_params(model_config)
    sets = [CODDataset(os.path.join(d, 'train'), params, True, device_preprocess) for d in dataset_dirs if os.path.exists(os.path.join(d, 'train'))]
    if not sets:
        raise ValueError("No valid training datasets found")
    full = ConcatDataset(sets)
    if val_ratio <= 0:
        return _loader(full, batch_size, True, num_workers, rank, world, drop_last), None
    n_train = int((1 - val_ratio) * len(full))
    tr, va = torch.utils.data.random_split(full, [n_train, len(full) - n_train], generator=torch.Generator().manual_seed(42))
    return _loader(tr, batch_size, True, num_workers, rank, world, drop_last), _loader(va, batch_size, False, num_workers, rank, world, False)


def get_test_loaders(dataset_dirs: List[str], model_config: Dict, batch_size: int = 16, num_workers: int = 4,
                     device_preprocess: bool = False) -> Dict[str, DataLoader]:
    params = _processor_params(model_config)
    loaders = {}
    for d in dataset_dirs:
        p = os.path.join(d, 'test')
        if os.path.exists(p):
            loaders[os.path.basename(d)] = _loader(CODDataset(p, params, False, device_preprocess), batch_size, False, num_workers)
    if not loaders:
        raise ValueError("No valid test datasets found")
    return loaders


class DeviceBatcher:
    """uint8 HWC images (host) -> normalised model input [B,3,S,S] float32 on the device.  Images go through two pinned staging buffers
    (double buffered) on a dedicated copy stream, then ONE batched preprocess launch; `submit` returns immediately, `result` makes the
    compute stream wait for the copy + kernel -- so the upload of batch k+1 overlaps the step on batch k."""

    def __init__(self, size, mean, std, device, max_bytes: int = 64 << 20):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.stage = [torch.empty(max_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(max_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.copied = [None, None]      # event after the last H2D copy out of each staging buffer
        self.turn = 0

    def submit(self, images_u8: List[torch.Tensor]):
        """Queues upload + preprocess of one batch; returns the handle `result` takes (two batches may be in flight: the one being trained
        on and the one prefetched behind it)."""
        from .. import ops
        k = self.turn
        self.turn ^= 1
        sizes = [(int(t.shape[0]), int(t.shape[1])) for t in images_u8]
        total = sum(h * w * 3 for h, w in sizes)
        if total > self.stage[k].numel():
            raise ValueError(f"batch of {total} image bytes exceeds the {self.stage[k].numel()}-byte staging buffer")
        if self.copied[k] is not None:
            self.copied[k].synchronize()            # the host must not overwrite a staging buffer whose upload is still queued
        off, offs = 0, []
        for t, (h, w) in zip(images_u8, sizes):
            n = h * w * 3
            self.stage[k][off:off + n].copy_(t.reshape(-1))
            offs.append(off)
            off += (n + 255) // 256 * 256
        with torch.cuda.stream(self.stream):
            self.dev[k][:off].copy_(self.stage[k][:off], non_blocking=True)
            self.copied[k] = torch.cuda.Event()
            self.copied[k].record(self.stream)
            out = ops.preprocess_batch(self.dev[k], offs, sizes, self.size, self.mean, self.std)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def result(self, handle) -> torch.Tensor:
        out, ev = handle
        torch.cuda.current_stream(self.device).wait_event(ev)
        out.record_stream(torch.cuda.current_stream(self.device))
        return out


def prefetch(loader, batcher: Optional[DeviceBatcher], device) -> Iterator[Dict]:
    """Iterates `loader` one batch ahead: while the caller trains on batch k, batch k+1 is already being uploaded (pinned host memory ->
    device on a copy stream) and, in device_preprocess mode, resized / normalised by the batched HIP kernel."""
    copy = torch.cuda.Stream(device=device)

    def stage(b):
        b = dict(b)
        with torch.cuda.stream(copy):
            if 'images_u8' in b:
                b['_images_pending'] = batcher.submit(b.pop('images_u8'))
            else:
                b['images'] = b['images'].to(device, non_blocking=True)
            for k in ('masks', 'edges'):
                if k in b:
                    b[k] = [t.to(device, non_blocking=True) for t in b[k]]
            ev = torch.cuda.Event()
            ev.record(copy)
        return b, ev

    def finish(b, ev):
        cur_stream = torch.cuda.current_stream(device)
        cur_stream.wait_event(ev)
        pending = b.pop('_images_pending', None)
        if pending is not None:
            b['images'] = batcher.result(pending)
        else:
            b['images'].record_stream(cur_stream)
        # the tensors were allocated on the copy stream's pool but are read on the compute stream: tell the allocator, or their blocks
        # could be handed to the upload of batch k+2 while kernels reading batch k are still queued
        for k in ('masks', 'edges'):
            for t in b.get(k, ()):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur_stream)
        return b

    it = iter(loader)
    try:
        cur = stage(next(it))
    except StopIteration:
        return
    for nxt in it:
        nxt = stage(nxt)
        yield finish(*cur)
        cur = nxt
    yield finish(*cur)
