#!/usr/bin/env python3
"""Golden vectors for the training-step tail, captured from the imported reference (build
container only): the LinearWarmupCosineAnnealingLR sequences of MoCE-IR-main/src/utils/schedulers.py for the two
configurations train.py uses and one with a non-zero floor, stepped past max_epochs.  Writes tests/golden/schedule_lr.npz.

FFTLoss (MoCE-IR-main/src/utils/loss_utils.py:139-152) is captured too (tests/golden/fft_loss.npz): that module imports
torchvision, torchvision.models.vgg19 and pytorch_msssim at module level, none installed here and none touched by FFTLoss, so the
capture puts empty stand-in modules into sys.modules for the import (the treatment moce_ir.py's unused fvcore import got), then
runs the reference class itself on seeded inputs: values and the gradient with respect to the prediction."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")


OUT = os.path.join(ROOT, "tests", "golden")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def capture_fft_loss():
    import types
    for name in ("torchvision", "torchvision.models", "pytorch_msssim"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.models"].vgg19 = None           # `from torchvision.models import vgg19`: only VGG losses call it
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    lu = load("MoCE-IR-main/src/utils/loss_utils.py", "ref_loss_utils")
    out = {}
    cases = {"a": (2, 3, 16, 16, 1.0, torch.float64), "b": (1, 3, 24, 40, 0.1, torch.float64), "c": (2, 3, 32, 32, 1.0, torch.float32),
             "odd": (1, 3, 15, 21, 0.5, torch.float64)}
    for name, (b, c, h, w, lw, dt) in cases.items():
        g = torch.Generator().manual_seed(len(name) * 31 + h)
        pred = torch.rand((b, c, h, w), generator=g, dtype=torch.float64).to(dt).requires_grad_(True)
        target = torch.rand((b, c, h, w), generator=g, dtype=torch.float64).to(dt)
        loss = lu.FFTLoss(loss_weight=lw)(pred, target)
        loss.backward()
        out[name + "_args"] = np.asarray([b, c, h, w, lw, 64 if dt == torch.float64 else 32], dtype=np.float64)
        out[name + "_pred"] = pred.detach().double().numpy()
        out[name + "_target"] = target.double().numpy()
        out[name + "_loss"] = np.asarray(float(loss.detach()), dtype=np.float64)
        out[name + "_dpred"] = pred.grad.double().numpy()
    np.savez_compressed(os.path.join(OUT, "fft_loss.npz"), **out)
    print("fft_loss:", {k: float(v) for k, v in out.items() if k.endswith("_loss")})


def main():
    capture_fft_loss()
    sch = load("MoCE-IR-main/src/utils/schedulers.py", "ref_schedulers")
    cases = {"train": (2e-4, 15, 150, 0.0, 0.0, 320), "finetune": (2e-4, 1, 40, 0.0, 0.0, 60),
             "floor": (1e-3, 5, 30, 1e-5, 1e-6, 70)}
    out = {}
    for name, (base, warm, mx, start, eta, steps) in cases.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base)
        s = sch.LinearWarmupCosineAnnealingLR(optimizer=opt, warmup_epochs=warm, max_epochs=mx, warmup_start_lr=start, eta_min=eta)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(steps):
            opt.step()
            s.step()
            lrs.append(opt.param_groups[0]["lr"])
        out[name + "_args"] = np.asarray([base, warm, mx, start, eta, steps], dtype=np.float64)
        out[name + "_lrs"] = np.asarray(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "schedule_lr.npz"), **out)
    print("schedule_lr:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
