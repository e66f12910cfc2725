"""Training-step tail: learning-rate schedule and frequency-domain loss."""
import os

import numpy as np
import pytest
import torch

from oracle import train_tail_ref as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "schedule_lr.npz"))
    for name in ("train", "finetune", "floor"):
        base, warm, mx, start, eta, steps = z[name + "_args"]
        yield name, float(base), int(warm), int(mx), float(start), float(eta), int(steps), z[name + "_lrs"]


def test_oracle_schedule_matches_reference_golden():
    """oracle restatement vs the sequence the imported reference scheduler produced (bit for bit: same recursion)."""
    for name, base, warm, mx, start, eta, steps, lrs in _cases():
        got = np.asarray(T.warmup_cosine_lrs(base, warm, mx, steps, start, eta))
        assert got.shape == lrs.shape
        assert np.array_equal(got, lrs), (name, np.abs(got - lrs).max())


class _FakeTrainer:
    def __init__(self, lr):
        self.lr = lr


@pytest.mark.parametrize("driver", ["trainer", "optimizer"])
def test_schedule_drop_in_matches_reference_golden(driver):
    """image_restoration_amd.schedule.LinearWarmupCosineAnnealingLR (reference constructor + step()/get_last_lr()) driving a
    FlatTrainer-like object and a torch optimizer; past max_epochs too (the periodic continuation)."""
    from image_restoration_amd.schedule import LinearWarmupCosineAnnealingLR
    for name, base, warm, mx, start, eta, steps, lrs in _cases():
        if driver == "trainer":
            target = _FakeTrainer(base)
            read = lambda: target.lr
        else:
            target = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=base)
            read = lambda: target.param_groups[0]["lr"]
        s = LinearWarmupCosineAnnealingLR(optimizer=target, warmup_epochs=warm, max_epochs=mx, warmup_start_lr=start, eta_min=eta)
        got = [read()]
        for _ in range(steps):
            s.step()
            got.append(read())
        assert np.array_equal(np.asarray(got), lrs), name
        assert s.get_last_lr() == [got[-1]]
        # resume: a fresh schedule loaded from the state continues the same sequence
        st = s.state_dict()
        t2 = _FakeTrainer(base)
        s2 = LinearWarmupCosineAnnealingLR(t2, warm, mx, start, eta)
        s2.load_state_dict(st)
        s.step(); s2.step()
        assert t2.lr == read()


def _fft_loss_cases():
    z = np.load(os.path.join(GOLD, "fft_loss.npz"))
    for name in ("a", "b", "c", "odd"):
        b, c, h, w, lw, bits = z[name + "_args"]
        yield name, float(lw), int(bits), torch.from_numpy(z[name + "_pred"]), torch.from_numpy(z[name + "_target"]), \
            float(z[name + "_loss"]), torch.from_numpy(z[name + "_dpred"])


def test_oracle_fft_loss_against_reference_golden():
    """The restatement against what the reference's FFTLoss class produced (tools/capture_golden_f3.py): value and gradient."""
    for name, lw, bits, pred, target, loss, dpred in _fft_loss_cases():
        dt = torch.float64 if bits == 64 else torch.float32
        p = pred.to(dt).requires_grad_(True)
        out = T.fft_loss(p, target.to(dt), lw)
        out.backward()
        tol = 1e-12 if bits == 64 else 1e-5
        assert abs(out.item() - loss) <= tol * max(1.0, abs(loss)), name
        assert (p.grad.double() - dpred).abs().max() <= tol * max(1.0, float(dpred.abs().max())), name


@pytest.mark.gpu
def test_fft_loss_on_gpu_against_reference_golden():
    """losses.FFTLoss (rocFFT + the native L1 reduction) against the reference-captured values; the odd-sized case too."""
    from image_restoration_amd.losses import FFTLoss
    dev = torch.device("cuda:0")
    for name, lw, bits, pred, target, loss, dpred in _fft_loss_cases():
        p = pred.float().to(dev).requires_grad_(True)
        out = FFTLoss(loss_weight=lw)(p, target.float().to(dev))
        out.backward()
        assert abs(out.item() - loss) <= 1e-4 * max(1.0, abs(loss)), name
        assert (p.grad.double().cpu() - dpred).norm() <= 1e-4 * float(dpred.norm()), name


def test_oracle_fft_loss_against_direct_dft():
    """the restatement against a dense-DFT evaluation in fp64 (independent of torch.fft)."""
    g = torch.Generator().manual_seed(7)
    pred = torch.rand((2, 3, 8, 12), generator=g, dtype=torch.float64)
    target = torch.rand((2, 3, 8, 12), generator=g, dtype=torch.float64)
    H, W = 8, 12
    ky, y = torch.arange(H, dtype=torch.float64)[:, None], torch.arange(H, dtype=torch.float64)[None, :]
    kx, x = torch.arange(W // 2 + 1, dtype=torch.float64)[:, None], torch.arange(W, dtype=torch.float64)[None, :]
    Fy = torch.exp(-2j * torch.pi * ky * y / H)
    Fx = torch.exp(-2j * torch.pi * kx * x / W)
    d = (pred - target).to(torch.complex128)
    D = torch.einsum("ky,bcyx,lx->bckl", Fy, d, Fx)
    ref = 0.25 * torch.cat([D.real.abs().reshape(-1), D.imag.abs().reshape(-1)]).mean()
    assert abs(T.fft_loss(pred, target, 0.25).item() - ref.item()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_losses_on_gpu_against_oracle(dtype):
    """L1Loss / FFTLoss modules (native mi_l1_loss reduction) vs the oracle, values and gradients."""
    from image_restoration_amd.losses import FFTLoss, L1Loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    pred = torch.rand((2, 3, 64, 96), generator=g).to(dtype)
    target = torch.rand((2, 3, 64, 96), generator=g).to(dtype)
    pr = pred.double().requires_grad_(True)
    ref_l1 = (pr - target.double()).abs().mean()
    ref_fft = T.fft_loss(pr, target.double(), 0.1)
    (ref_l1 + ref_fft).backward()
    p = pred.to(dev).requires_grad_(True)
    l1 = L1Loss()(p, target.to(dev))
    fl = FFTLoss(loss_weight=0.1)(p, target.to(dev))
    (l1 + fl).backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert abs(l1.item() - ref_l1.item()) < tol * max(1.0, abs(ref_l1.item()))
    assert abs(fl.item() - ref_fft.item()) < tol * max(1.0, abs(ref_fft.item()))
    num = (p.grad.double().cpu() - pr.grad).norm() / pr.grad.norm()
    assert num < (1e-4 if dtype == torch.float32 else 2e-2)
    with pytest.raises(RuntimeError):
        L1Loss()(pred, target)          # CPU tensors are refused: no fallback
