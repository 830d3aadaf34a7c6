"""Input step before the hot path on the GPU.

The reference decodes with PIL and runs `ToTensor -> Normalize -> RandomHorizontalFlip -> RandomVerticalFlip ->
RandomResizedCrop(input_size, scale=(0.25, 1.0), bicubic, antialias)` per sample on the CPU workers (util/datasets.py:120-136),
then copies fp32 batches to the device (main_pretrain.py:376-394).  Here the workers only decode; the uint8 pixels go to the
device through pinned, double-buffered staging (4x fewer PCIe bytes than fp32) and one HIP kernel (`csmae_augment_u8`) does
the whole transform chain.  The random decisions are drawn on the host in torchvision's order, so a seeded run makes the same
flips and boxes as the reference's transform would for the same image sizes.
"""
from __future__ import annotations

import math
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

FMOW_RGB_MEAN = (0.43392888, 0.43578541, 0.40744025)   # util/datasets.py:167-168
FMOW_RGB_STD = (0.19828456, 0.19250111, 0.19454683)


def sample_transform_params(H: int, W: int, scale=(0.25, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), p_flip=0.5) -> Tuple[int, ...]:
    """(H, W, i, j, h, w, hflip, vflip) for one image, consuming the global CPU torch RNG like the reference's Compose does:
    RandomHorizontalFlip (`torch.rand(1) < p`), RandomVerticalFlip, then RandomResizedCrop.get_params (torchvision 0.15.1: up to
    ten (area, log-ratio) proposals with `uniform_`, `randint` for the corner, centre-crop fallback clamped to the ratio range)."""
    hflip = int(torch.rand(1).item() < p_flip)
    vflip = int(torch.rand(1).item() < p_flip)
    return (H, W) + resized_crop_box(H, W, scale, ratio) + (hflip, vflip)


def resized_crop_box(H: int, W: int, scale=(0.25, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Tuple[int, int, int, int]:
    """(i, j, h, w) of torchvision 0.15.1 `RandomResizedCrop.get_params`, drawn from the global CPU torch RNG."""
    area = float(H * W)
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = math.exp(torch.empty(1).uniform_(lo, hi).item())
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= W and 0 < h <= H:
            i = torch.randint(0, H - h + 1, size=(1,)).item()
            j = torch.randint(0, W - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(W) / float(H)
    if in_ratio < ratio[0]:
        w = W; h = int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        h = H; w = int(round(h * ratio[1]))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


class GpuAugment:
    """uint8 HWC images (list of tensors / arrays of possibly different sizes) -> normalised fp32 [N, C, S, S] on `device`."""

    def __init__(self, input_size: int, mean: Sequence[float] = FMOW_RGB_MEAN, std: Sequence[float] = FMOW_RGB_STD, scale=(0.25, 1.0),
                 device="cuda", slots: int = 2):
        from csmae_hip import load
        load()  # fail loudly when the HIP library is missing: there is no CPU path behind this class
        self.S, self.scale, self.device = int(input_size), tuple(scale), torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.inv_std = 1.0 / torch.tensor(std, dtype=torch.float32, device=self.device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots, self._k = [None] * slots, 0

    def _slot(self, nbytes, N, C):
        k = self._k % len(self._slots)
        self._k += 1
        s = self._slots[k]
        if s is None or s["host"].numel() < nbytes or s["meta_host"].shape[0] < N or s["out"].shape[0] < N or s["out"].shape[1] != C:
            cap, capn = int(nbytes * 1.25) + 4096, max(N, 8)
            s = dict(host=torch.empty(cap, dtype=torch.uint8).pin_memory(), dev=torch.empty(cap, dtype=torch.uint8, device=self.device),
                     meta_host=torch.empty(capn, 8, dtype=torch.int32).pin_memory(), meta_dev=torch.empty(capn, 8, dtype=torch.int32, device=self.device),
                     out=torch.empty(capn, C, self.S, self.S, dtype=torch.float32, device=self.device), staged=None)
            self._slots[k] = s
        if s["staged"] is not None:
            s["staged"].synchronize()  # the H2D copy that last read this slot's pinned buffer
        return s

    def stage(self, images, params: Optional[List[Tuple[int, ...]]] = None):
        """Everything for one batch on the copy stream: H2D of the uint8 pixels, then the transform kernel into the slot's output
        buffer — it overlaps the training step of the previous batch.  `images` is either a list of uint8 HWC tensors / arrays
        (packed here) or a `PackedBatch` built by `collate_uint8` inside the loader workers (then the main process only does one
        large copy into pinned memory).  Returns a handle for `finish`."""
        from csmae_hip import ops
        if not isinstance(images, PackedBatch):
            images = pack_uint8([torch.as_tensor(im) for im in images])
        N, Hmax, Wmax, C = images.data.shape
        nbytes = images.data.numel()
        s = self._slot(nbytes, N, C)
        s["host"][:nbytes].view(N, Hmax, Wmax, C).copy_(images.data)
        if params is None:
            params = [sample_transform_params(int(h), int(w), self.scale) for h, w in images.sizes.tolist()]
        s["meta_host"][:N] = torch.tensor(params, dtype=torch.int32)
        # the slot's device buffers were last read by the training step two batches ago, enqueued on the current stream
        self.copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.copy_stream):
            s["dev"][:nbytes].copy_(s["host"][:nbytes], non_blocking=True)
            s["meta_dev"][:N].copy_(s["meta_host"][:N], non_blocking=True)
            s["staged"] = torch.cuda.Event()
            s["staged"].record(self.copy_stream)
            src = s["dev"][:nbytes].view(N, Hmax, Wmax, C)
            ops.augment_u8(src, s["meta_dev"][:N], self.mean, self.inv_std, s["out"][:N], st=self.copy_stream.cuda_stream)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return dict(slot=s, N=N, ready=ready)

    def finish(self, h) -> torch.Tensor:
        """The batch as fp32 [N, C, S, S]; valid until the slot comes round again (`slots` batches later)."""
        torch.cuda.current_stream().wait_event(h["ready"])
        return h["slot"]["out"][: h["N"]]

    def __call__(self, images, params=None):
        return self.finish(self.stage(images, params))


class PrefetchLoader:
    """Wraps an iterable of (list of uint8 HWC images, labels): batch k+1 is packed and copied while batch k trains."""

    def __init__(self, batches: Iterable, augment: GpuAugment):
        self.batches, self.augment = batches, augment

    def __iter__(self):
        it = iter(self.batches)
        try:
            images, labels = next(it)
        except StopIteration:
            return
        pending = (self.augment.stage(images), labels)
        for images, labels in it:
            nxt = (self.augment.stage(images), labels)
            yield self.augment.finish(pending[0]), pending[1]
            pending = nxt
        yield self.augment.finish(pending[0]), pending[1]

    def __len__(self):
        return len(self.batches)


class CsvImageDataset(torch.utils.data.Dataset):
    """fMoW-RGB style CSV (column 0 = label, column 1 = image path, util/datasets.py:161-206): returns the decoded uint8 HWC image;
    no transform runs on the CPU."""

    def __init__(self, csv_path: str):
        import pandas as pd
        self.base = os.path.dirname(csv_path)
        info = pd.read_csv(csv_path, header=0)
        self.paths, self.labels = list(info.iloc[:, 1]), list(info.iloc[:, 0])

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        import numpy as np
        from PIL import Image
        p = self.paths[i]
        p = p if os.path.isabs(p) else os.path.join(self.base, p)
        with Image.open(p) as im:
            arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(arr.copy()), self.labels[i]


class PackedBatch:
    """Decoded images of one batch in one uint8 tensor [N, Hmax, Wmax, C] (image n in the top-left corner) + their sizes [N, 2]."""

    def __init__(self, data: torch.Tensor, sizes: torch.Tensor):
        self.data, self.sizes = data, sizes


def pack_uint8(images) -> PackedBatch:
    N, C = len(images), int(images[0].shape[2])
    Hmax, Wmax = max(int(im.shape[0]) for im in images), max(int(im.shape[1]) for im in images)
    data = torch.zeros(N, Hmax, Wmax, C, dtype=torch.uint8)
    for n, im in enumerate(images):
        data[n, : im.shape[0], : im.shape[1]] = im
    return PackedBatch(data, torch.tensor([[int(im.shape[0]), int(im.shape[1])] for im in images], dtype=torch.int32))


def collate_uint8(samples):
    """DataLoader collate_fn: runs in the worker processes, so decoding AND packing are parallel; the main process copies once."""
    return pack_uint8([s[0] for s in samples]), torch.as_tensor([s[1] for s in samples])
