"""World-size-2 gloo test of the data-parallel trainer (image_restoration_amd/trainer.py): two ranks on half batches
must end with the same parameters as one process on the whole batch with torch.optim.AdamW — which checks the flat
parameter/gradient buffers, the stage-bucketed all-reduce launched from the backward hooks (overlap=True) and after
backward (overlap=False), the folding of autograd-delivered gradients (the AdamW arithmetic
itself is injected by this test: the product's update is a HIP kernel and refuses CPU tensors).  CPU only; the blocks of
the stand-in network are plain torch layers because the HIP modules refuse CPU tensors by design."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class TinyNet(nn.Module):
    """Three parameterised stages plus a parameter-free one and a skip connection, like the U-Net's top level."""

    def __init__(self):
        super().__init__()
        self.embed = nn.Conv2d(3, 8, 3, padding=1)
        self.enc = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), nn.GELU(), nn.Conv2d(8, 8, 1))
        self.pool = nn.Identity()
        self.dec = nn.Sequential(nn.Conv2d(16, 8, 1), nn.GELU())
        self.out = nn.Conv2d(8, 3, 3, padding=1, bias=False)

    def forward(self, x):
        e = self.embed(x)
        h = self.pool(self.enc(e))
        d = self.dec(torch.cat([h, e], 1))
        return self.out(d) + x


def _host_adamw(tr, scale):
    """AdamW on the trainer's flat host buffers: stands in for the fused HIP kernel (mi_adamw_step) so that the
    bucketing / all-reduce bookkeeping can run on CPU tensors; the product itself has no CPU update."""
    import math
    g = tr.flat_g * scale
    b1, b2 = tr.betas
    tr.flat_p.mul_(1.0 - tr.lr * tr.wd)
    tr.flat_m.mul_(b1).add_(g, alpha=1 - b1)
    tr.flat_v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** tr.step_count, math.sqrt(1 - b2 ** tr.step_count)
    tr.flat_p.addcdiv_(tr.flat_m, tr.flat_v.sqrt() / bc2 + tr.eps, value=-tr.lr / bc1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, overlap, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_restoration_amd.trainer import FlatTrainer
        torch.manual_seed(0)
        model = TinyNet()
        sharded = overlap == "sharded"
        tr = FlatTrainer(model, lr=1e-2, overlap=bool(overlap) and not sharded, host_update=_host_adamw, shard_optimizer=sharded)
        if sharded:     # moments cover this rank's slice only; the flat buffers are padded to world equal shards
            assert tr.flat_m.numel() == tr.shard and tr.flat_p.numel() == tr.shard * world and not tr.overlap
            sd = tr.state_dict()
            assert sd["shard"] == (rank, world)
            tr.load_state_dict(sd)
        # embed, enc.0, enc.2, dec.0, out: nn.Sequential stages are bucketed per block
        assert tr.world == world and [n for n, *_ in tr.stages] == ["embed", "enc.0", "enc.2", "dec.0", "out"]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 3, 8, 8, generator=g)
        y = torch.randn(4, 3, 8, 8, generator=g)
        xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
        for _ in range(3):
            tr.zero_grad()
            loss = (model(xs) - ys).abs().mean()
            loss.backward()
            tr.reduce_gradients()
            tr.optimizer_step()
        if rank == 0:
            ret["params"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
            ret["stages"] = [(n, lo, hi) for n, _, lo, hi in tr.stages]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False, "sharded"])
def test_two_rank_training_matches_single_process(overlap):
    """overlap True / False: all-reduce from the backward hooks / after backward; "sharded": reduce-scatter, AdamW on each rank's
    slice of the flat parameters, all-gather (FlatTrainer(shard_optimizer=True))."""
    torch.manual_seed(0)
    ref = TinyNet()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randn(4, 3, 8, 8, generator=g)
    for _ in range(3):
        opt.zero_grad()
        (ref(x) - y).abs().mean().backward()
        opt.step()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), overlap, ret), nprocs=2, join=True)
    got = ret["params"]
    for k, v in ref.state_dict().items():
        assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-6), k
    # flat ranges of the stages are disjoint and ordered by registration
    spans = sorted((lo, hi) for _, lo, hi in ret["stages"])
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))


class TinyMoE(nn.Module):
    """Two 'experts' of which a rank may route to only one (MoCE top-1 routing leaves experts grad-less in a step; the
    reference runs DDP with find_unused_parameters=True, MoCE-IR-main/src/train.py:131)."""

    def __init__(self):
        super().__init__()
        self.embed = nn.Conv2d(3, 4, 1)
        self.e0 = nn.Conv2d(4, 4, 1)
        self.e1 = nn.Conv2d(4, 4, 1)
        self.out = nn.Conv2d(4, 3, 1)

    def forward(self, x, which):
        h = self.embed(x)
        h = self.e0(h) if which == 0 else self.e1(h)
        return self.out(h)


def _worker_moe(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_restoration_amd.trainer import FlatTrainer
        torch.manual_seed(0)
        model = TinyMoE()
        tr = FlatTrainer(model, lr=1e-2, overlap=True, host_update=_host_adamw)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 3, 4, 4, generator=g)
        y = torch.randn(4, 3, 4, 4, generator=g)
        # rank r routes its half batch to expert r: each expert's bucket is all zeros on the other rank
        # two micro-batches per step: the first one accumulates locally inside no_sync()
        for _ in range(2):
            tr.zero_grad()
            with tr.no_sync():
                ((model(x[rank * 2:rank * 2 + 1], rank) - y[rank * 2:rank * 2 + 1]).abs().mean() * 0.5).backward()
            ((model(x[rank * 2 + 1:rank * 2 + 2], rank) - y[rank * 2 + 1:rank * 2 + 2]).abs().mean() * 0.5).backward()
            tr.reduce_gradients()
            tr.optimizer_step()
        sd = tr.state_dict()
        tr2_m = TinyMoE()
        tr2 = FlatTrainer(tr2_m, lr=5e-2, overlap=False, host_update=_host_adamw)
        tr2_m.load_state_dict(model.state_dict())
        tr2.load_state_dict(sd)
        assert tr2.step_count == 2 and tr2.lr == 1e-2 and torch.equal(tr2.flat_m, tr.flat_m)
        # a second forward without no_sync() must be refused instead of silently corrupting a reduced bucket
        tr.zero_grad()
        model(x[:1], rank).sum().backward()
        try:
            model(x[:1], rank)
            refused = False
        except RuntimeError as e:
            refused = "no_sync" in str(e)
        if rank == 0:
            ret["params"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
            ret["refused"] = refused
    finally:
        dist.destroy_process_group()


def test_unrouted_expert_zero_bucket_and_accumulation():
    """MoCE under data parallelism: an expert a rank did not route to contributes an all-zero bucket; with no_sync()
    micro-batches the result equals one process on the whole batch (mean over 4 samples)."""
    torch.manual_seed(0)
    ref = TinyMoE()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 4, 4, generator=g)
    y = torch.randn(4, 3, 4, 4, generator=g)
    for _ in range(2):
        opt.zero_grad()
        loss = 0
        for i in range(4):
            loss = loss + (ref(x[i:i + 1], i // 2) - y[i:i + 1]).abs().mean() / 4
        loss.backward()
        opt.step()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_moe, args=(2, _free_port(), ret), nprocs=2, join=True)
    for k, v in ref.state_dict().items():
        assert torch.allclose(ret["params"][k], v, rtol=1e-5, atol=1e-6), k
    assert ret["refused"]


class TinyContainers(nn.Module):
    """the shapes the trainer's 'did not run this step' shortcut was unsound for - a ModuleList as the LAST
    child (its forward hook can never fire: MoCE-IR's enc / dec), a module whose weights are used functionally (its forward is
    never called: Restormer's reduce_chan / output / up2_1), and a bare-parameter container."""

    def __init__(self):
        super().__init__()
        self.embed = nn.Conv2d(3, 4, 1)
        self.mid = nn.Conv2d(4, 4, 1)                                  # used functionally (weights read, forward never called)
        self.groups = nn.ModuleList([nn.ModuleList([nn.Conv2d(4, 4, 1), nn.Conv2d(4, 4, 1)]), nn.Conv2d(4, 3, 1)])

    def forward(self, x):
        from image_restoration_amd.restormer import _fire_forward_hooks
        h = self.embed(x)
        y = torch.nn.functional.conv2d(h, self.mid.weight, self.mid.bias)
        _fire_forward_hooks(self.mid, (h,), y)                         # what the package's functional uses do
        for blk in self.groups[0]:
            y = blk(y)
        return self.groups[1](y)


def _worker_containers(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_restoration_amd.trainer import FlatTrainer
        torch.manual_seed(0)
        model = TinyContainers()
        tr = FlatTrainer(model, lr=1e-2, overlap=True, host_update=_host_adamw)
        names = [n for n, *_ in tr.stages]
        g = torch.Generator().manual_seed(5)
        x = torch.randn(4, 3, 4, 4, generator=g)
        y = torch.randn(4, 3, 4, 4, generator=g)
        launched = []
        orig = tr._launch_reduce

        def spy(idx):
            # at the moment a bucket is handed to the collective, every gradient of its stage must already be in it
            _, child, lo, hi = tr.stages[idx]
            tr._fold_autograd_grads(child)
            launched.append((names[idx], float(tr.flat_g[lo:hi].abs().sum())))
            orig(idx)
        tr._launch_reduce = spy
        for _ in range(2):
            tr.zero_grad()
            (model(x[rank * 2:rank * 2 + 2]) - y[rank * 2:rank * 2 + 2]).abs().mean().backward()
            tr.reduce_gradients()
            tr.optimizer_step()
        if rank == 0:
            ret["params"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
            ret["names"] = names
            ret["launched"] = launched
    finally:
        dist.destroy_process_group()


def test_containers_and_functional_modules_are_not_reduced_early():
    torch.manual_seed(0)
    ref = TinyContainers()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 4, 4, generator=g)
    y = torch.randn(4, 3, 4, 4, generator=g)
    for _ in range(2):
        opt.zero_grad()
        (ref(x) - y).abs().mean().backward()
        opt.step()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_containers, args=(2, _free_port(), ret), nprocs=2, join=True)
    # the ModuleLists were expanded into their elements: every stage is a module with a forward of its own
    assert ret["names"] == ["embed", "mid", "groups.0.0", "groups.0.1", "groups.1"]
    # no bucket went out empty (a stage reduced before its gradients were written would show a zero sum here)
    assert all(s > 0 for _, s in ret["launched"]), ret["launched"]
    for k, v in ref.state_dict().items():
        assert torch.allclose(ret["params"][k], v, rtol=1e-5, atol=1e-6), k


def test_cosine_warmup_schedule_closed_form():
    """LinearWarmupCosineAnnealingLR(warmup 15, max 150) closed form (MoCE-IR-main/src/utils/schedulers.py:332-346):
    hand-computed points."""
    import math
    from image_restoration_amd.trainer import cosine_warmup_lr
    assert cosine_warmup_lr(0, 2e-4) == 0.0
    assert abs(cosine_warmup_lr(7, 2e-4) - 7 * 2e-4 / 14) < 1e-15
    assert abs(cosine_warmup_lr(14, 2e-4) - 2e-4) < 1e-15
    assert abs(cosine_warmup_lr(15, 2e-4) - 2e-4) < 1e-15
    mid = 15 + (150 - 15) / 2
    assert abs(cosine_warmup_lr(mid, 2e-4) - 1e-4) < 1e-12
    assert abs(cosine_warmup_lr(150, 2e-4)) < 1e-15
    assert abs(cosine_warmup_lr(60, 2e-4) - 0.5 * 2e-4 * (1 + math.cos(math.pi * 45 / 135))) < 1e-15


def test_optimizer_step_has_no_cpu_path():
    from image_restoration_amd.trainer import FlatTrainer
    tr = FlatTrainer(TinyNet(), lr=1e-2)
    with pytest.raises(RuntimeError, match="no CPU optimizer path"):
        tr.optimizer_step()
