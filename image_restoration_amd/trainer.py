"""Data-parallel training step for the drop-in networks: flat fp32 parameter / gradient / Adam buffers,
stage-bucketed gradient all-reduce (RCCL over xGMI on the GPU, any torch.distributed backend in tests)
overlapped with backward, and one fused AdamW kernel over the flat buffer.

Counterpart of the reference harness ``MoCE-IR-main/src/train.py:26-148`` (Lightning DDP + AdamW(lr=2e-4),
L1 loss): one process per GPU, pure data parallelism, gradients averaged over ranks.  Differences by design
(MI355X-first): parameters live in ONE flat buffer (one optimizer launch; a handful of large collectives
instead of 25 MB autograd buckets), and the blocks accumulate weight gradients straight into the flat gradient
buffer (``param.main_grad``), so no per-parameter autograd accumulation kernels run.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops

ALIGN = 64  # elements: every parameter starts on a 256-byte boundary of the flat buffers


class FlatTrainer:
    def __init__(self, model: nn.Module, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, process_group=None, overlap: bool = True, host_update=None,
                 pack_cache: bool = True):
        self.model = model
        # host_update(trainer, scale): test hook that stands in for the fused AdamW kernel when the gradient-bucketing /
        # all-reduce bookkeeping is exercised on CPU tensors over gloo.  The product has no CPU update: without the hook,
        # optimizer_step on CPU tensors raises.
        self._host_update = host_update
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "model has no trainable parameters"
        dev = params[0].device
        for p in params:
            assert p.dtype == torch.float32 and p.device == dev, "parameters must be fp32 on one device"
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.offsets: Dict[int, Tuple[int, int]] = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.main_grad = self.flat_g[o:o + n].view(p.shape)
                p.grad = None
                self.offsets[id(p)] = (o, n)
        self.params = params
        self.total = total
        # stages = top-level children that own parameters; their flat ranges are contiguous by construction
        self.stages: List[Tuple[str, nn.Module, int, int]] = []
        for name, child in model.named_children():
            ps = [p for p in child.parameters() if p.requires_grad]
            if not ps:
                continue
            lo = min(self.offsets[id(p)][0] for p in ps)
            hi = max(self.offsets[id(p)][0] + (self.offsets[id(p)][1] + ALIGN - 1) // ALIGN * ALIGN for p in ps)
            self.stages.append((name, child, lo, hi))
        self.overlap = overlap and self.world > 1
        self._exec_order: List[int] = []
        self._reduced: set = set()
        self._works = []
        self._comm_stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.world > 1) else None
        if self.overlap:
            for idx, (_, child, _, _) in enumerate(self.stages):
                child.register_forward_hook(self._make_fwd_hook(idx))
        self.dev_scalars = torch.zeros(3, dtype=torch.float32, device=dev) if dev.type == "cuda" else None
        # The trainer is the only writer of the parameters, so it can let the library keep the packed (bf16, LDS-image)
        # copies of all 1x1 weights across calls and refresh them once per optimizer step (mi_pw_cache_*): 4 bytes of
        # cache per parameter covers both orientations of every matrix in either activation dtype, plus tile padding.
        self._pack_cache = bool(pack_cache) and dev.type == "cuda"
        if self._pack_cache:
            ops.pw_cache_enable(int(total) * 8 + (4 << 20), dev, self.flat_p)

    # ------------------------------------------------------------------ gradient bookkeeping
    def zero_grad(self) -> None:
        self.flat_g.zero_()
        self._exec_order.clear()
        self._reduced.clear()
        self._works.clear()

    def _fold_autograd_grads(self, module: nn.Module) -> None:
        """Glue layers that still run as PyTorch ops deliver .grad through autograd: add it into main_grad."""
        for p in module.parameters():
            if p.grad is not None:
                p.main_grad.add_(p.grad)
                p.grad = None

    def _make_fwd_hook(self, idx: int):
        def hook(module, inputs, output):
            if not torch.is_grad_enabled() or not isinstance(output, torch.Tensor) or not output.requires_grad:
                return
            self._exec_order.append(idx)
            pos = len(self._exec_order) - 1

            def on_grad(_grad):
                # grad w.r.t. this stage's output is complete => every stage executed after it has finished backward
                for later in self._exec_order[pos + 1:]:
                    self._launch_reduce(later)
                return None
            output.register_hook(on_grad)
        return hook

    def _launch_reduce(self, idx: int) -> None:
        if idx in self._reduced or self.world == 1:
            return
        self._reduced.add(idx)
        _, child, lo, hi = self.stages[idx]
        self._fold_autograd_grads(child)
        buf = self.flat_g[lo:hi]
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def reduce_gradients(self) -> None:
        """Call after backward: folds autograd-delivered grads, all-reduces whatever is not yet in flight, waits."""
        if self.world == 1:
            self._fold_autograd_grads(self.model)
            return
        for idx in range(len(self.stages)):
            self._launch_reduce(idx)
        for w in self._works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._works.clear()

    # ------------------------------------------------------------------ optimizer
    def set_step_scalars(self, step: int, lr: Optional[float] = None) -> None:
        """Refresh the device-side {lr, bias corrections} (call before replaying a captured step)."""
        lr = self.lr if lr is None else lr
        vals = torch.tensor([lr, 1.0 - self.betas[0] ** step, math.sqrt(1.0 - self.betas[1] ** step)],
                            dtype=torch.float32)
        self.dev_scalars.copy_(vals, non_blocking=True)

    def optimizer_step(self, use_dev_scalars: bool = False) -> None:
        self.step_count += 1
        scale = 1.0 / self.world
        if self.flat_p.is_cuda:
            ops.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.step_count, self.betas,
                           self.eps, self.wd, scale, self.dev_scalars if use_dev_scalars else None)
            if self._pack_cache:
                ops.pw_cache_refresh()   # the weights just changed: re-pack every 1x1 weight image in one launch
        elif self._host_update is not None:
            self._host_update(self, scale)
        else:
            raise RuntimeError("FlatTrainer.optimizer_step: parameters are not on an MI355X (no CPU optimizer path)")


def cosine_warmup_lr(epoch: int, base_lr: float, warmup_epochs: int = 15, max_epochs: int = 150,
                     warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> float:
    """Closed form of LinearWarmupCosineAnnealingLR (MoCE-IR-main/src/utils/schedulers.py:332-346;
    train.py:84-88 uses warmup_epochs=15, max_epochs=150)."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / max(1, warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - warmup_epochs) /
                                                               (max_epochs - warmup_epochs)))
