#!/usr/bin/env python3
"""Golden vectors for the training-step tail, captured from the imported reference (build
container only): the LinearWarmupCosineAnnealingLR sequences of MoCE-IR-main/src/utils/schedulers.py for the two
configurations train.py uses and one with a non-zero floor, stepped past max_epochs.  Writes tests/golden/schedule_lr.npz.

FFTLoss is not captured: MoCE-IR-main/src/utils/loss_utils.py imports torchvision and pytorch_msssim at module level and
neither is installed here, so that module does not import (oracle/train_tail_ref.py says so: parity unpinned for FFTLoss)."""
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


def main():
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
