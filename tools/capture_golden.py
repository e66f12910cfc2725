#!/usr/bin/env python3
"""Capture golden vectors from the imported reference (runs in the build container only).

Imports ``/root/reference/Restormer.py`` (pure torch + einops), fills parameters
from the version-stable seeded generator shared with the oracle
(``oracle.restormer_ref.make_state``), runs CPU forwards/backwards and writes small
``tests/golden/*.npz`` fixtures: inputs are regenerated from the seed by the
tests; the fixtures store outputs and gradients only.  No reference source or
bytecode enters this repository; the GPU box never sees ``/root/reference``.

Usage:  python tools/capture_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REF)

from oracle import restormer_ref as R  # noqa: E402
from oracle.fixtures import pack, seeded_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    """Tensors are stored compacted (strided subset + digests); scalars/strings as they are."""
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            pack(k, v, out)
        else:
            out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def grads_of(module, x, seed):
    """fwd + bwd with a seeded cotangent; returns (y, dx, {param: grad})."""
    x = x.clone().requires_grad_(True)
    y = module(x)
    cot = seeded_input(tuple(y.shape), seed + 1000, y.dtype)
    y.backward(cot)
    return y.detach(), x.grad.detach(), {k: p.grad.detach() for k, p in module.named_parameters()}


def main():
    import Restormer as ref  # the reference module, imported from REF
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # (1) LayerNorm, both flavours
    for kind in ("WithBias", "BiasFree"):
        for c in (16, 48, 96):
            m = ref.LayerNorm(c, kind).double()
            rng = np.random.default_rng(10 + c)
            sd = {"body.weight": R.seeded_tensor(rng, (c,), "ln_w", torch.float64)}
            if kind == "WithBias":
                sd["body.bias"] = R.seeded_tensor(rng, (c,), "ln_b", torch.float64)
            m.load_state_dict(sd)
            x = seeded_input((1, c, 8, 8), 100 + c, torch.float64) * 1.5 + 0.3
            y, dx, g = grads_of(m, x, 100 + c)
            save(f"ln_{kind}_{c}", y=y, dx=dx, **{"g_" + k: v for k, v in g.items()})

    # (2) FeedForward / (3) Attention / (4) TransformerBlock
    cases = [("c48h1", 48, 1, (2, 48, 16, 16)), ("c16h1", 16, 1, (2, 16, 16, 16)),
             ("c96h2", 96, 2, (2, 96, 16, 16)), ("c96h1", 96, 1, (2, 96, 16, 16)),
             ("c48h1_64", 48, 1, (1, 48, 64, 64))]
    for tag, c, heads, shape in cases:
        for bias in (False, True):
            if bias and tag != "c48h1":
                continue
            btag = tag + ("_bias" if bias else "")
            sd = R.make_block_state(c, heads, 2.66, bias, "WithBias", seed=7 + c + heads, dtype=torch.float64)
            x = seeded_input(shape, 200 + c + heads, torch.float64)

            ffn = ref.FeedForward(c, 2.66, bias).double()
            ffn.load_state_dict(R.sub_state(sd, "ffn."))
            y, dx, g = grads_of(ffn, x, 300)
            save(f"ffn_{btag}", y=y, dx=dx, **{"g_" + k: v for k, v in g.items()})

            att = ref.Attention(c, heads, bias).double()
            att.load_state_dict(R.sub_state(sd, "attn."))
            y, dx, g = grads_of(att, x, 400)
            save(f"attn_{btag}", y=y, dx=dx, **{"g_" + k: v for k, v in g.items()})

            blk = ref.TransformerBlock(c, heads, 2.66, bias, "WithBias").double()
            blk.load_state_dict(sd)
            y, dx, g = grads_of(blk, x, 500)
            save(f"block_{btag}", y=y, dx=dx, **{"g_" + k: v for k, v in g.items()})

    # BiasFree block
    sd = R.make_block_state(48, 1, 2.66, False, "BiasFree", seed=77, dtype=torch.float64)
    blk = ref.TransformerBlock(48, 1, 2.66, False, "BiasFree").double()
    blk.load_state_dict(sd)
    x = seeded_input((2, 48, 16, 16), 277, torch.float64)
    y, dx, g = grads_of(blk, x, 500)
    save("block_c48h1_biasfree", y=y, dx=dx, **{"g_" + k: v for k, v in g.items()})

    # (5) Restormer-tiny 1x3x128x128 (config C1): fp32 forward as the reference runs it,
    #     plus an fp64 forward (the parity target) and the L1-loss gradient norms.
    cfg = R.RESTORMER_TINY
    kw = {k: cfg[k] for k in ("inp_channels", "out_channels", "dim", "num_blocks", "num_refinement_blocks",
                              "heads", "ffn_expansion_factor", "bias", "LayerNorm_type")}
    net = ref.Restormer(**kw)
    sd32 = R.make_restormer_state(cfg, seed=1)
    assert list(sd32.keys()) == list(net.state_dict().keys()), "state_dict key order differs from reference"
    net.load_state_dict(sd32)
    rng = np.random.default_rng(1234)
    clean = torch.from_numpy(rng.random((1, 3, 128, 128))).to(torch.float32)
    noisy = R.degrade_sigma(clean, 25.0, seed=4321)
    with torch.no_grad():
        y32 = net(noisy)
    net64 = ref.Restormer(**kw).double()
    net64.load_state_dict({k: v.double() for k, v in sd32.items()})
    out = net64(noisy.double())
    loss = (out - clean.double()).abs().mean()
    loss.backward()
    gn = {k: float(p.grad.norm()) for k, p in net64.named_parameters()}
    keys = sorted(gn)
    save("restormer_tiny_128", y32=y32[:, :, ::4, ::4], y64=out.detach()[:, :, ::4, ::4],
         y64_mean=float(out.mean()), y64_abs_mean=float(out.abs().mean()), loss=float(loss),
         psnr_in=R.psnr(noisy, clean), psnr_out=R.psnr(out, clean),
         grad_norm_keys=np.array(keys), grad_norms=np.array([gn[k] for k in keys]))

    # Restormer base: key list + parameter count only (no weights)
    base = ref.Restormer()
    shapes = R.restormer_param_shapes(R.RESTORMER_BASE)
    ref_shapes = {k: tuple(v.shape) for k, v in base.state_dict().items()}
    assert shapes == ref_shapes and list(shapes) == list(ref_shapes), "base key/shape mismatch"
    save("restormer_base_keys", keys=np.array(list(ref_shapes)), n_params=sum(p.numel() for p in base.parameters()))


if __name__ == "__main__":
    main()
