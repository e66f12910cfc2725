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
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops

ALIGN = 64  # elements: every parameter starts on a 256-byte boundary of the flat buffers


class FlatTrainer:
    def __init__(self, model: nn.Module, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, process_group=None, overlap: bool = True, host_update=None,
                 pack_cache: bool = True, bucket_blocks: bool = True, shard_optimizer: bool = False):
        """shard_optimizer: reduce-scatter the flat gradient, run AdamW on this rank's 1/world slice of the parameters (the two
        moment buffers shrink to that slice), all-gather the updated parameters - instead of all-reduce + replicated AdamW.
        One collective pair per step after backward (no per-stage overlap); for models whose optimizer state matters."""
        self.model = model
        # host_update(trainer, scale): test hook that stands in for the fused AdamW kernel when the gradient-bucketing /
        # all-reduce bookkeeping is exercised on CPU tensors over gloo.  The product has no CPU update: without the hook,
        # optimizer_step on CPU tensors raises.
        self._host_update = host_update
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pg = process_group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if inited else 1
        # MI_FORCE_COMM=1: run every collective even in a one-rank group (rehearsal of the RCCL code path on a one-GPU box)
        self._comm = self.world > 1 or (inited and os.environ.get("MI_FORCE_COMM") == "1")
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
        self.sharded = bool(shard_optimizer) and self._comm
        self.rank = dist.get_rank(process_group) if self._comm else 0
        if self.sharded:                                   # equal, aligned shards: pad the flat buffers at the end
            self.shard = (total + self.world * ALIGN - 1) // (self.world * ALIGN) * ALIGN
            total = self.shard * self.world
        else:
            self.shard = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros_like(self.flat_m)
        self._g_shard = torch.zeros(self.shard, dtype=torch.float32, device=dev) if self.sharded else None
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
        # reduction units ("stages"): the top-level children that own parameters, with nn.Sequential stages split into
        # their blocks (a 256^2 stage's four blocks take ~10 ms of backward each: per-block buckets let the first reduce
        # start that much earlier and leave only the patch embedding behind the last one); flat ranges are contiguous
        # by construction (parameters() order = registration order)
        self.stages: List[Tuple[str, nn.Module, int, int]] = []

        def units_of(name, mod):
            # a ModuleList has no forward: its forward hook would never fire, so its ELEMENTS are the stages (recursively:
            # MoCE-IR's enc / dec are ModuleLists of ModuleLists); an nn.Sequential of blocks becomes one stage per block
            if isinstance(mod, nn.ModuleList):
                out = []
                for i, sub in enumerate(mod):
                    out += units_of(f"{name}.{i}", sub)
                return out
            if bucket_blocks and isinstance(mod, nn.Sequential) and len(mod) > 1:
                return [(f"{name}.{i}", sub) for i, sub in enumerate(mod)]
            return [(name, mod)]
        for name, child in model.named_children():
            for uname, unit in units_of(name, child):
                ps = [p for p in unit.parameters() if p.requires_grad]
                if not ps:
                    continue
                lo = min(self.offsets[id(p)][0] for p in ps)
                hi = max(self.offsets[id(p)][0] + (self.offsets[id(p)][1] + ALIGN - 1) // ALIGN * ALIGN for p in ps)
                self.stages.append((uname, unit, lo, hi))
        self._param_lists: dict = {}
        self.overlap = overlap and self._comm and not self.sharded
        self._exec_order: List[int] = []
        self._reduced: set = set()
        self._works = []
        self._comm_stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self._comm) else None
        # a stage may be declared ready early ("did not run this step") only if it would have fired a forward hook had it
        # run: true for modules with a forward of their own; containers that only hold parameters are left to
        # reduce_gradients()
        self._hookable = [type(child).forward is not nn.Module.forward and not isinstance(child, (nn.ModuleList, nn.ModuleDict,
                                                                                                nn.ParameterList, nn.ParameterDict))
                          for _, child, _, _ in self.stages]
        if self.overlap:
            for idx, (_, child, _, _) in enumerate(self.stages):
                child.register_forward_hook(self._make_fwd_hook(idx))
        self.dev_scalars = torch.zeros(3, dtype=torch.float32, device=dev) if dev.type == "cuda" else None
        # The trainer is the only writer of the parameters, so it can let the library keep the packed (bf16, LDS-image)
        # copies of all 1x1 weights across calls and refresh them once per optimizer step (mi_pw_cache_*): 4 bytes of
        # cache per parameter covers both orientations of every matrix in either activation dtype, plus tile padding.
        # Every weight gradient ends in a fixed-order sum of partial rows; with an arena lent to the library those ~500 small
        # launches per step are recorded during backward and run as one table-driven launch before the gradients are used
        # (reduce_gradients / each bucket's all-reduce).  MI_DEFER_MB sizes the arena (0 disables; overflow falls back to
        # immediate sums).  Restormer base at bs 32 x 256^2 records 4.8 GB of partial rows per backward (tools/debug_defer_hw.py;
        # the per-image rows of the weight-gradient Grams are most of it): the 3 GiB of rounds 3-4 overflowed there and the
        # attention and LayerNorm sums of the early levels, recorded last, fell back to ~120 immediate launches per step.
        defer_mb = int(ops.env("MI_DEFER_MB") or 6144)
        self._defer_token = ops.deferred_begin(defer_mb << 20, dev) if (dev.type == "cuda" and defer_mb > 0) else None
        self._pack_cache = bool(pack_cache) and dev.type == "cuda"
        self._sync = True            # False inside no_sync(): micro-batches accumulate locally, nothing is reduced
        self._seen_fwd: set = set()
        self._ready: set = set()
        self._next = len(self.stages) - 1
        self._bwd_started = False
        if self._pack_cache:
            self._cache_token = ops.pw_cache_enable(int(total) * 8 + (4 << 20), dev, self.flat_p)
            # Anything that writes the parameters other than optimizer_step (load_state_dict to resume or to evaluate a
            # checkpoint, an EMA copy-back, a manual re-init) must not leave the GEMMs on stale packed images: torch bumps
            # a parameter's version counter on every in-place write, the fused AdamW kernel (a raw pointer write) does
            # not, so a change of the summed counters seen at the next forward means "somebody else wrote": re-pack.
            self._p_version = self._weights_version()
            model.register_load_state_dict_post_hook(lambda *_: self.weights_changed())
            model.register_forward_pre_hook(lambda *_: self._check_weights())

    # ------------------------------------------------------------------ packed-weight cache safety
    def _weights_version(self) -> int:
        """Changes whenever torch wrote a parameter in place: every parameter is a view into the flat buffer made through
        ``.data``, which does NOT share the buffer's version counter, so the parameters' own counters are summed (plus the
        buffer's, for writes to it directly).  The fused AdamW kernel writes through raw pointers and bumps nothing: the
        optimizer step refreshes the cache itself."""
        return self.flat_p._version + sum(p._version for p in self.params)

    def weights_changed(self) -> None:
        """Tell the trainer the parameters were written outside optimizer_step (called automatically after
        load_state_dict and whenever the flat buffer's version counter moved)."""
        ops.bump_weights_epoch()      # caches of weight-derived data outside the library (fused-kernel packs, fp8 scales)
        if self._pack_cache:
            ops.pw_cache_refresh()
            self._p_version = self._weights_version()

    def _check_weights(self) -> None:
        if self._pack_cache and self._weights_version() != self._p_version:
            self.weights_changed()

    def flush_deferred(self) -> None:
        """Run the parameter-gradient sums recorded since the last flush (no-op without a deferral context)."""
        if self._defer_token is not None:
            ops.deferred_flush()

    def close(self) -> None:
        """Detach the library's packed-weight cache from this trainer's buffers.  The cache is process-global: it is switched
        off only if it still belongs to this trainer (a trainer or PackedWeights made later keeps its own)."""
        if getattr(self, "_defer_token", None) is not None:
            ops.deferred_end(self._defer_token)
            self._defer_token = None
        if self._pack_cache:
            ops.pw_cache_release(getattr(self, "_cache_token", None))
            self._cache_token = None
            self._pack_cache = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ checkpointing (optimizer state; the model's own
    # state_dict carries the parameters: Lightning's checkpoint holds both, MoCE-IR-main/src/train.py:107-116,137-148)
    def state_dict(self) -> dict:
        return {"exp_avg": self.flat_m.clone(), "exp_avg_sq": self.flat_v.clone(), "step": self.step_count, "lr": self.lr,
                "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "numel": int(self.total),
                "shard": (self.rank, self.world) if self.sharded else None}

    def load_state_dict(self, sd: dict) -> None:
        if int(sd["numel"]) != int(self.total):
            raise ValueError(f"optimizer state is for {sd['numel']} flat elements, this trainer has {self.total}")
        want = (self.rank, self.world) if self.sharded else None
        if sd.get("shard") != want:
            raise ValueError(f"optimizer state was saved for shard {sd.get('shard')}, this trainer is {want}")
        self.flat_m.copy_(sd["exp_avg"])
        self.flat_v.copy_(sd["exp_avg_sq"])
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        self.betas, self.eps, self.wd = tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
        self.weights_changed()

    def no_sync(self):
        """Context manager for gradient accumulation (Lightning's accumulate_grad_batches, MoCE-IR-main/src/train.py:133):
        backward passes inside it only accumulate into the local flat gradient; run the LAST micro-batch outside it (or
        just call reduce_gradients afterwards) to all-reduce everything once."""
        trainer = self

        class _NoSync:
            def __enter__(self_inner):
                trainer._sync = False

            def __exit__(self_inner, *exc):
                trainer._sync = True
                return False
        return _NoSync()

    # ------------------------------------------------------------------ gradient bookkeeping
    def zero_grad(self) -> None:
        if self._defer_token is not None:
            ops.deferred_flush()              # (nothing should be pending; a stale job must not land in the zeroed buffer)
            ops.deferred_record(True)         # this step's backward may defer its parameter-gradient sums
        self.flat_g.zero_()
        self._exec_order.clear()
        self._reduced.clear()
        self._works.clear()
        self._seen_fwd.clear()
        self._ready.clear()
        self._next = len(self.stages) - 1
        self._bwd_started = False

    def _fold_autograd_grads(self, module: nn.Module) -> None:
        """Glue layers that still run as PyTorch ops deliver .grad through autograd: add it into main_grad.
        (The parameter list of a module is cached: walking the module tree of MoCE-IR base every step cost 5 ms of host
        time in a launch-bound step.)"""
        ps = self._param_lists.get(id(module))
        if ps is None:
            ps = self._param_lists[id(module)] = [p for p in module.parameters() if p.requires_grad]
        for p in ps:
            if p.grad is not None:
                p.main_grad.add_(p.grad)
                p.grad = None

    def _make_fwd_hook(self, idx: int):
        def hook(module, inputs, output):
            if not torch.is_grad_enabled() or not isinstance(output, torch.Tensor) or not output.requires_grad:
                return
            if not self._sync:
                return               # accumulating: no per-stage reduction for this micro-batch
            if idx in self._seen_fwd:
                # a second forward before zero_grad() would add into a bucket that may already have been reduced
                raise RuntimeError("FlatTrainer: stage ran twice between zero_grad() calls with overlapped reduction on; "
                                   "wrap all but the last micro-batch in trainer.no_sync()")
            self._seen_fwd.add(idx)
            self._exec_order.append(idx)
            pos = len(self._exec_order) - 1

            def on_grad(_grad):
                # grad w.r.t. this stage's output is complete => every stage executed after it has finished backward.
                # (Holds for the chain-shaped top level of Restormer / MoCE-IR / AdaIR, where a later stage consumes an
                # earlier one's output; skip connections only ADD consumers that ran later still.)
                if not self._bwd_started:
                    # Stages whose forward hook did not fire this step did not run (an optional branch): nothing will ever be
                    # written into their slice, so they are ready from the start.  This is sound only because every stage that
                    # DOES run fires its hook: ModuleList containers are expanded into their elements (units_of above) and the
                    # modules this package uses functionally (glue convs, reduce_chan, up2_1) run their hooks explicitly
                    # (restormer._fire_forward_hooks).  A stage that cannot fire one (a parameter used outside any module
                    # forward) is left to reduce_gradients(): see _hookable.
                    self._bwd_started = True
                    for i in range(len(self.stages)):
                        if i not in self._seen_fwd and self._hookable[i]:
                            self._mark_ready(i)
                for later in self._exec_order[pos + 1:]:
                    self._mark_ready(later)
                return None
            output.register_hook(on_grad)
        return hook

    def _mark_ready(self, idx: int) -> None:
        """Collectives must be issued in ONE order on every rank, whatever order the stages finished in (ranks may route
        to different experts): launch strictly by descending stage index, each as soon as it and all later ones are ready."""
        self._ready.add(idx)
        while self._next >= 0 and self._next in self._ready:
            self._launch_reduce(self._next)
            self._next -= 1

    def _launch_reduce(self, idx: int) -> None:
        if idx in self._reduced or not self._comm:
            return
        self._reduced.add(idx)
        _, child, lo, hi = self.stages[idx]
        self.flush_deferred()           # the bucket's gradients must be final before they go out
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
        self.grads_ready()
        if not self._comm:
            self._fold_autograd_grads(self.model)
            return
        if self.sharded:
            self._fold_autograd_grads(self.model)
            lo = self.rank * self.shard
            if dist.get_backend(self.pg) == "gloo":        # gloo has no reduce-scatter: same result by all-reduce + slice
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
                self._g_shard.copy_(self.flat_g[lo:lo + self.shard])
            else:
                dist.reduce_scatter_tensor(self._g_shard, self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
            return
        for idx in range(len(self.stages) - 1, -1, -1):
            self._mark_ready(idx)
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

    def grads_ready(self) -> None:
        """Make ``flat_g`` / every ``main_grad`` final: run the parameter-gradient sums the backward recorded (they are deferred
        into one launch, ops.deferred_*) and stop recording.  reduce_gradients() and optimizer_step() call it; anything else that
        reads gradients between backward and the optimizer step (clipping by norm, logging) must call it first."""
        if self._defer_token is not None:
            ops.deferred_flush()
            ops.deferred_record(False)

    def capture_step(self, step_fn, warmup: int = 2):
        """Capture ``step_fn`` (a whole training step ending in ``optimizer_step(use_dev_scalars=True)``) into a HIP graph and
        return it; ``replay_step(graph)`` runs it.  The warm-up steps run on a SIDE stream: a parameter whose gradient arrives
        through autograd's AccumulateGrad (MoCE-IR's router gates, the embedding MLP) binds that node to the stream of its
        first backward, and a node bound to the legacy default stream takes hipStreamEndCapture down.  Pending deferred sums
        are flushed first; while the capture runs nothing is deferred (the library sums at once on a capturing stream)."""
        assert self.flat_p.is_cuda, "capture_step needs the parameters on the GPU"
        self.grads_ready()
        side = torch.cuda.Stream(device=self.flat_p.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.set_step_scalars(self.step_count + 1)
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.grads_ready()
        for p in self.params:
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        self.set_step_scalars(self.step_count + 1)
        with torch.cuda.graph(graph):
            step_fn()
        return graph

    def replay_step(self, graph) -> None:
        """One more step of a captured graph: refresh the device-side {lr, bias corrections}, then replay."""
        self.set_step_scalars(self.step_count + 1)
        self.step_count += 1
        ops.bump_weights_epoch()
        graph.replay()

    def optimizer_step(self, use_dev_scalars: bool = False) -> None:
        self.grads_ready()            # a loop that skips reduce_gradients() (one GPU) must not step on incomplete gradients
        self.step_count += 1
        ops.bump_weights_epoch()      # the fused AdamW kernel writes the parameters without bumping any version counter
        scale = 1.0 / self.world
        if self.sharded:
            lo = self.rank * self.shard
            p_shard = self.flat_p[lo:lo + self.shard]
            if self.flat_p.is_cuda:
                ops.adamw_step(p_shard, self._g_shard, self.flat_m, self.flat_v, self.lr, self.step_count, self.betas, self.eps,
                               self.wd, scale, self.dev_scalars if use_dev_scalars else None)
            elif self._host_update is not None:
                import types
                self._host_update(types.SimpleNamespace(flat_p=p_shard, flat_g=self._g_shard, flat_m=self.flat_m,
                                                        flat_v=self.flat_v, betas=self.betas, lr=self.lr, wd=self.wd,
                                                        eps=self.eps, step_count=self.step_count), scale)
            else:
                raise RuntimeError("FlatTrainer.optimizer_step: parameters are not on an MI355X (no CPU optimizer path)")
            if dist.get_backend(self.pg) == "gloo":
                parts = [torch.empty_like(p_shard) for _ in range(self.world)]
                dist.all_gather(parts, p_shard.clone(), group=self.pg)
                for r, part in enumerate(parts):
                    self.flat_p[r * self.shard:(r + 1) * self.shard].copy_(part)
            else:
                dist.all_gather_into_tensor(self.flat_p, p_shard, group=self.pg)     # in place: each rank's slice is its input
            if self._pack_cache and self.flat_p.is_cuda:
                ops.pw_cache_refresh()
                self._p_version = self._weights_version()
            return
        if self.flat_p.is_cuda:
            ops.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.step_count, self.betas,
                           self.eps, self.wd, scale, self.dev_scalars if use_dev_scalars else None)
            if self._pack_cache:
                ops.pw_cache_refresh()   # the weights just changed: re-pack every 1x1 weight image in one launch
                self._p_version = self._weights_version()
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
