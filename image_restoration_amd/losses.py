"""Loss terms of the reference's training step on the native L1 kernel.

``L1Loss`` stands in for ``nn.L1Loss()`` (MoCE-IR-main/src/train.py:51,54) and ``FFTLoss`` for
``MoCE-IR-main/src/utils/loss_utils.py:139-152`` (``--loss_type fft``): the mean absolute difference of the real and
imaginary parts of ``rfft2`` of prediction and target.  Both reduce with ``mi_l1_loss`` (one pass that also produces the
gradient); the transform itself is ``torch.fft.rfft2`` - rocFFT is a library call here, plumbing like device memory -
and its backward is autograd's.  GPU tensors only: the product has no CPU path."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import ops


class _L1MeanFn(torch.autograd.Function):
    """mean|a - b| with the gradient taken in the same kernel pass."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        a, b = a.contiguous(), b.contiguous()
        loss, da = ops.l1_loss(a, b, want_grad=True)
        ctx.save_for_backward(da)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        (da,) = ctx.saved_tensors
        ga = da * g.to(da.dtype)
        return ga, (-ga if ctx.needs_input_grad[1] else None)


class L1Loss(nn.Module):
    """``nn.L1Loss(reduction='mean')`` on the native kernel."""

    def __init__(self, reduction: str = "mean") -> None:
        super().__init__()
        if reduction != "mean":
            raise ValueError("only reduction='mean' is implemented (the reference uses the default)")

    def forward(self, pred: Tensor, target: Tensor) -> Tensor:
        return _L1MeanFn.apply(pred, target.to(pred.dtype))


class FFTLoss(nn.Module):
    """``loss_weight * L1(stack(re, im)(rfft2(pred)), stack(re, im)(rfft2(target)))`` - loss_utils.py:139-152."""

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean") -> None:
        super().__init__()
        if reduction != "mean":
            raise ValueError("only reduction='mean' is implemented (the reference passes the default)")
        self.loss_weight = loss_weight

    def forward(self, pred: Tensor, target: Tensor) -> Tensor:
        # view_as_real lays (re, im) out innermost, exactly what the reference's torch.stack(..., dim=-1) builds
        pf = torch.view_as_real(torch.fft.rfft2(pred.float()))
        tf = torch.view_as_real(torch.fft.rfft2(target.float()))
        return self.loss_weight * _L1MeanFn.apply(pf, tf)
