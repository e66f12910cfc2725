"""Helpers shared by ``tools/capture_golden.py`` and the tests (TEST INFRASTRUCTURE).

Golden fixtures hold outputs only; inputs are regenerated from seeds.  Large arrays
are stored as a deterministic strided subset plus whole-array digests (sum, L2 norm)
so that each ``tests/golden/*.npz`` stays small.
"""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MAX_ELEMS = 4096


def seeded_input(shape, seed, dtype=torch.float32):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape)).to(dtype)


def compact(a, max_elems: int = MAX_ELEMS):
    """-> dict(sub=strided subset, sum=..., l2=...) of a tensor/array (fp64 digests)."""
    a = np.asarray(a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    flat = a.reshape(-1)
    stride = max(1, -(-flat.size // max_elems))
    return {"sub": flat[::stride].astype(np.float32), "sum": np.float64(flat.sum()),
            "l2": np.float64(np.sqrt((flat * flat).sum()))}


def pack(prefix: str, a, out: dict, max_elems: int = MAX_ELEMS):
    for k, v in compact(a, max_elems).items():
        out[f"{prefix}.{k}"] = v


def check(prefix: str, got, gold, rtol: float, atol_scale: float = 1.0, what: str = ""):
    """Compare ``got`` with the packed golden entry: subset within rtol*max|gold| and digests."""
    c = compact(got)
    ref_sub = np.asarray(gold[f"{prefix}.sub"], dtype=np.float64)
    scale = max(float(np.abs(ref_sub).max()), 1e-30)
    err = float(np.abs(c["sub"].astype(np.float64) - ref_sub).max()) / scale
    assert err <= rtol * atol_scale, f"{what}{prefix}: max rel err {err:.3e} > {rtol * atol_scale:.1e}"
    l2 = float(gold[f"{prefix}.l2"])
    assert abs(c["l2"] - l2) <= 10 * rtol * atol_scale * max(l2, 1e-30), \
        f"{what}{prefix}: l2 {c['l2']:.6e} vs {l2:.6e}"
    return err


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
