"""PSNR / SSIM evaluation on the device, with the reference harness's conventions
(AdaIR-main/utils/val_utils.py:50-64 ``compute_psnr_ssim``; MoCE-IR-main/src/test.py:82-123): both images clipped to
[0, 1]; PSNR = 10 log10(1 / MSE) per image; SSIM = scikit-image's ``structural_similarity(data_range=1, multichannel)`` per
image (7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, 3-pixel border dropped); batch means returned."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib as L
from . import ops


def psnr_ssim_per_image(restored: torch.Tensor, clean: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (psnr [B], ssim [B]) as fp32 device tensors.  restored, clean: [B, C, H, W] float32 / bfloat16 on the GPU."""
    assert restored.shape == clean.shape and restored.dtype == clean.dtype
    restored, clean = restored.contiguous(), clean.contiguous()
    ops._gpu(restored, clean)
    B, Cc, H, W = restored.shape
    psnr = torch.empty(B, dtype=torch.float32, device=restored.device)
    ssim = torch.empty_like(psnr)
    ws = ops._blob(L.lib().mi_psnr_ssim_workspace(B, Cc, H, W), restored.device)
    L.check(L.lib().mi_psnr_ssim(restored.data_ptr(), clean.data_ptr(), psnr.data_ptr(), ssim.data_ptr(), B, Cc, H, W,
                                 ops._dt(restored), ws.data_ptr(), ops._stream()), "psnr_ssim")
    return psnr, ssim


def compute_psnr_ssim(recoverd: torch.Tensor, clean: torch.Tensor):
    """The reference's signature (val_utils.py:50): -> (mean PSNR, mean SSIM, batch size)."""
    psnr, ssim = psnr_ssim_per_image(recoverd, clean)
    return float(psnr.mean()), float(ssim.mean()), int(recoverd.shape[0])
