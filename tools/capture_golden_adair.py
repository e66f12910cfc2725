#!/usr/bin/env python3
"""Golden vectors for AdaIR's frequency modules and the assembled network, captured from the imported reference
(AdaIR-main/net/model.py; build container only).  Parameters come from the seeded generator shared with the oracle; fixtures
hold outputs and gradients only (compacted).  Writes tests/golden/adair_fre_*.npz, adair_tiny_train.npz, adair_keys.npz."""
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

from oracle import restormer_ref as R  # noqa: E402
from oracle.fixtures import pack, seeded_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
F64 = torch.float64


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            pack(k, v, out)
        else:
            out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def load_adair():
    spec = importlib.util.spec_from_file_location("adair_model", os.path.join(REF, "AdaIR-main", "net", "model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def fill(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(R.make_state(shapes, seed, F64), strict=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ad = load_adair()

    # (1) FreModule, small feature map (16 x 16 < 128: the low-frequency rectangle is empty, as in every training step)
    for tag, dim, heads, img_hw, hw, B in (("c32", 32, 2, 64, 16, 2), ("c16_big", 16, 2, 768, 384, 2)):
        m = ad.FreModule(dim, heads, False).double()
        fill(m, 900 + dim)
        img = seeded_input((B, 3, img_hw, img_hw), 910 + dim, F64)
        y = seeded_input((B, dim, hw, hw), 911 + dim, F64).requires_grad_(True)
        out = m(img, y)
        cot = seeded_input(tuple(out.shape), 912 + dim, F64)
        out.backward(cot)
        # the rectangle the reference used, recomputed exactly as its fft() does (pins the oracle's mask arithmetic)
        with torch.no_grad():
            feat = m.conv1(torch.nn.functional.interpolate(img, (hw, hw), mode="bilinear"))
            thr = m.rate_conv(torch.nn.functional.adaptive_avg_pool2d(feat, 1)).sigmoid()
            half = torch.stack(((hw // 128 * thr[:, 0, 0, 0]).int(), (hw // 128 * thr[:, 1, 0, 0]).int()), 1)
        print(tag, "half sizes", half.tolist())
        g = {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}
        save(f"adair_fre_{tag}", y=out.detach(), dy=y.grad.detach(), half=half.numpy().astype(np.int64),
             **{"g_" + k: v for k, v in g.items()})

    # (2) the assembled network, tiny widths, train-mode forward + L1 backward
    cfg = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 2], ffn_expansion_factor=2.66,
               bias=False, LayerNorm_type="WithBias", decoder=True)
    net = ad.AdaIR(**cfg).double()
    fill(net, 950)
    img = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(951), dtype=F64)
    tgt = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(952), dtype=F64)
    out = net(img)
    loss = (out - tgt).abs().mean()
    loss.backward()
    gn = {k: (float(p.grad.norm()) if p.grad is not None else -1.0) for k, p in net.named_parameters()}
    save("adair_tiny_train", y=out.detach(), loss=float(loss), grad_names=np.array(list(gn)), grad_norms=np.array(list(gn.values())))
    save("adair_keys", adair_base=np.array(list(ad.AdaIR().state_dict())), fre=np.array(list(ad.FreModule(48, 4, False).state_dict())))


if __name__ == "__main__":
    main()
