"""TEST INFRASTRUCTURE ONLY - CPU restatement of the training-step tail: the learning-rate
schedule and the frequency-domain loss.  Only tests/ may import this module.

warmup_cosine_lrs is pinned against the imported reference: tests/golden/schedule_lr.npz was written by
tools/capture_golden_f3.py from MoCE-IR-main/src/utils/schedulers.py:239-346 (tests/test_train_tail.py).
fft_loss is pinned too (round 4): tests/golden/fft_loss.npz holds losses and prediction gradients produced by the reference's own
FFTLoss class (MoCE-IR-main/src/utils/loss_utils.py:139-152), imported by tools/capture_golden_f3.py with empty stand-ins for
the module-level torchvision / pytorch_msssim imports that class never touches; tests/test_train_tail.py checks this
restatement (and the product's FFTLoss on the GPU) against it, besides the independent dense-DFT evaluation."""
from __future__ import annotations

import math
from typing import List

import torch


def warmup_cosine_lrs(base_lr: float, warmup_epochs: int, max_epochs: int, steps: int, warmup_start_lr: float = 0.0,
                      eta_min: float = 0.0) -> List[float]:
    """The sequence lr_0, lr_1, ... produced by constructing the scheduler and calling step() `steps` times
    (schedulers.py:296-330: the chainable form, every value derived from the one before)."""
    out: List[float] = []
    lr = base_lr
    span = max_epochs - warmup_epochs
    for t in range(steps + 1):
        if t == 0:
            lr = warmup_start_lr
        elif t < warmup_epochs:
            lr = lr + (base_lr - warmup_start_lr) / (warmup_epochs - 1)
        elif t == warmup_epochs:
            lr = base_lr
        elif (t - 1 - max_epochs) % (2 * span) == 0:
            lr = lr + (base_lr - eta_min) * (1 - math.cos(math.pi / span)) / 2
        else:
            lr = ((1 + math.cos(math.pi * (t - warmup_epochs) / span)) /
                  (1 + math.cos(math.pi * (t - warmup_epochs - 1) / span)) * (lr - eta_min) + eta_min)
        out.append(lr)
    return out


def fft_loss(pred: torch.Tensor, target: torch.Tensor, loss_weight: float = 1.0) -> torch.Tensor:
    """loss_utils.py:145-152: mean |.| over the stacked real and imaginary parts of rfft2(pred) - rfft2(target)."""
    d = torch.fft.rfft2(pred) - torch.fft.rfft2(target)
    return loss_weight * torch.cat([d.real.abs().reshape(-1), d.imag.abs().reshape(-1)]).mean()
