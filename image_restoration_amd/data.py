"""Device-side training samples: the denoise branch of the reference's ``AIOTrainDataset.__getitem__``
(MoCE-IR-main/src/data/dataset_utils.py:156-165) - crop to a multiple of 16, random P x P crop, random dihedral augmentation,
Gaussian noise of sigma 15 / 25 / 50 on the uint8 grid, ToTensor - as ONE kernel launch per batch over a device-resident pool
of decoded uint8 images (``mi_patch_batch``), writing the clean and degraded batches directly in the activation dtype.

Decoding image files (PIL) is outside the hot path and stays on the host: ``ImagePool.add`` takes decoded HWC uint8 arrays.
The random choices (sample, crop origin, augmentation mode, noise) are drawn with a torch.Generator so that a run is
reproducible; the reference draws them with ``random`` / ``numpy.random`` in its DataLoader workers."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops

SIGMA_OF_TASK = {"denoise_15": 15.0, "denoise_25": 25.0, "denoise_50": 50.0}     # degradation_utils.py:26-38


def crop_to_multiple(image: np.ndarray, base: int = 16) -> np.ndarray:
    """image_utils.crop_img: centre-crop H and W down to a multiple of ``base``."""
    h, w = image.shape[:2]
    ch, cw = h % base, w % base
    return image[ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2]


class ImagePool:
    """Decoded uint8 HWC images packed into one flat device buffer (+ offset / size tables)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._host: List[np.ndarray] = []
        self.pool = self.off = self.h = self.w = None

    def add(self, image: np.ndarray, base: int = 16) -> None:
        assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3, "decoded RGB uint8 HWC image expected"
        self._host.append(np.ascontiguousarray(crop_to_multiple(image, base)))
        self.pool = None

    def __len__(self):
        return len(self._host)

    def finalize(self) -> None:
        sizes = [im.size for im in self._host]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = np.concatenate([im.reshape(-1) for im in self._host])
        self.pool = torch.from_numpy(flat).to(self.device)
        self.off = torch.from_numpy(offs).to(self.device)
        self.h = torch.tensor([im.shape[0] for im in self._host], dtype=torch.int32, device=self.device)
        self.w = torch.tensor([im.shape[1] for im in self._host], dtype=torch.int32, device=self.device)
        self._hw = torch.tensor([[im.shape[0], im.shape[1]] for im in self._host], dtype=torch.int64)


def draw_batch_plan(pool: ImagePool, batch: int, patch: int, sigmas: Sequence[float], gen: torch.Generator):
    """Host-side random plan of one batch: (sample, top, left, mode, sigma) per element, all int32 / fp32 CPU tensors."""
    n = len(pool)
    sample = torch.randint(0, n, (batch,), generator=gen)
    hw = pool._hw[sample]
    assert int(hw.min()) >= patch, "every pooled image must be at least patch x patch"
    top = (torch.rand(batch, generator=gen) * (hw[:, 0] - patch + 1).float()).long().clamp_(min=0)
    left = (torch.rand(batch, generator=gen) * (hw[:, 1] - patch + 1).float()).long().clamp_(min=0)
    top = torch.minimum(top, hw[:, 0] - patch)
    left = torch.minimum(left, hw[:, 1] - patch)
    # random_augmentation draws random.randint(1, 7) (image_utils.py:182, both ends inclusive): the identity (mode 0) is never taken
    mode = torch.randint(1, 8, (batch,), generator=gen)
    sig = torch.tensor(list(sigmas), dtype=torch.float32)[torch.randint(0, len(sigmas), (batch,), generator=gen)]
    return sample.int(), top.int(), left.int(), mode.int(), sig


def patch_batch(pool: ImagePool, plan, patch: int, dtype=torch.bfloat16, noise: Optional[torch.Tensor] = None,
                gen: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (degraded, clean) [B, 3, P, P] on the device in ``dtype``: one launch for crop + augment + noise + ToTensor."""
    if pool.pool is None:
        pool.finalize()
    dev = pool.device
    sample, top, left, mode, sig = (t.to(dev) for t in plan)
    B = int(sample.numel())
    if noise is None:
        noise = torch.randn((B, 3, patch, patch), dtype=torch.float32, device=dev, generator=gen)
    clean = torch.empty((B, 3, patch, patch), dtype=dtype, device=dev)
    degraded = torch.empty_like(clean)
    L.check(L.lib().mi_patch_batch(pool.pool.data_ptr(), pool.off.data_ptr(), pool.h.data_ptr(), pool.w.data_ptr(),
                                   sample.data_ptr(), top.data_ptr(), left.data_ptr(), mode.data_ptr(), sig.data_ptr(),
                                   noise.contiguous().data_ptr(), clean.data_ptr(), degraded.data_ptr(), B, patch,
                                   ops._dt(clean), ops._stream()), "patch_batch")
    return degraded, clean
