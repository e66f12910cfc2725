"""Learning-rate schedule of the reference's training step.

``LinearWarmupCosineAnnealingLR`` keeps the reference's class name, constructor arguments and ``step()`` /
``get_last_lr()`` methods (MoCE-IR-main/src/utils/schedulers.py:239-346, used by MoCE-IR-main/src/train.py:82-88 with
warmup_epochs=15, max_epochs=150) but drives anything that exposes a learning rate: a ``FlatTrainer`` (``.lr``), or a
``torch.optim`` optimizer (``param_groups[i]["lr"]``).  Like the reference's ``step()`` without an epoch argument it
follows the *chainable* recursion - each value is computed from the previous one - so the floats agree with the reference
bit for bit (tests/test_schedule.py against tests/golden/schedule_lr.npz)."""
from __future__ import annotations

import math
from typing import List


class _Target:
    """Uniform view of the learning rates being driven."""

    def __init__(self, obj):
        self.obj = obj
        self.groups = getattr(obj, "param_groups", None)

    def get(self) -> List[float]:
        if self.groups is not None:
            return [float(g["lr"]) for g in self.groups]
        return [float(self.obj.lr)]

    def set(self, lrs: List[float]) -> None:
        if self.groups is not None:
            for g, lr in zip(self.groups, lrs):
                g["lr"] = lr
        else:
            self.obj.lr = lrs[0]


class LinearWarmupCosineAnnealingLR:
    """Linear warm-up from ``warmup_start_lr`` to the base rate over ``warmup_epochs`` steps, then cosine annealing to
    ``eta_min`` at ``max_epochs`` (and periodic beyond it)."""

    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                 eta_min: float = 0.0, last_epoch: int = -1) -> None:
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.warmup_start_lr = warmup_start_lr
        self.eta_min = eta_min
        self._target = _Target(optimizer)
        self.base_lrs = self._target.get()
        self.last_epoch = last_epoch
        self._last_lr = list(self.base_lrs)
        self.step()                          # torch's scheduler base class performs this initial step too

    def _next(self, current: List[float]) -> List[float]:
        t, w, m = self.last_epoch, self.warmup_epochs, self.max_epochs
        if t == 0:
            return [self.warmup_start_lr for _ in self.base_lrs]
        if t < w:
            return [lr + (base - self.warmup_start_lr) / (w - 1) for base, lr in zip(self.base_lrs, current)]
        if t == w:
            return list(self.base_lrs)
        span = m - w
        if (t - 1 - m) % (2 * span) == 0:    # restart point of the periodic continuation
            return [lr + (base - self.eta_min) * (1 - math.cos(math.pi / span)) / 2
                    for base, lr in zip(self.base_lrs, current)]
        num = 1 + math.cos(math.pi * (t - w) / span)
        den = 1 + math.cos(math.pi * (t - w - 1) / span)
        return [num / den * (lr - self.eta_min) + self.eta_min for lr in current]

    def step(self) -> None:
        self.last_epoch += 1
        lrs = self._next(self._target.get())
        self._target.set(lrs)
        self._last_lr = lrs

    def get_last_lr(self) -> List[float]:
        return list(self._last_lr)

    def state_dict(self) -> dict:
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "_last_lr": list(self._last_lr)}

    def load_state_dict(self, state: dict) -> None:
        self.last_epoch = int(state["last_epoch"])
        self.base_lrs = list(state["base_lrs"])
        self._last_lr = list(state["_last_lr"])
        self._target.set(self._last_lr)
