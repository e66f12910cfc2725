#!/usr/bin/env python3
"""Golden vectors for the MoCE-IR / AdaIR block pieces, captured from the imported reference (build container only).

``/root/reference/moce_ir.py`` imports ``fvcore`` (absent; used only by commented-out code): a two-name
stub module is registered before the import.  The router's N(0,1) draw is injected by patching ``torch.randn_like``
during the call so that fixtures are reproducible.  Writes tests/golden/moce_*.npz and adair_*.npz."""
from __future__ import annotations

import importlib.util
import os
import sys
import types

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


class injected_noise:
    """torch.randn_like -> a fixed seeded draw (moved to the argument's dtype/device)."""

    def __init__(self, seed):
        self.seed = seed

    def __enter__(self):
        self.orig = torch.randn_like
        seed = self.seed

        def fake(t, **kw):
            return seeded_input(tuple(t.shape), seed, torch.float64).to(t.dtype).to(t.device)
        torch.randn_like = fake

    def __exit__(self, *a):
        torch.randn_like = self.orig


def load_moce():
    stub = types.ModuleType("fvcore")
    nn_stub = types.ModuleType("fvcore.nn")
    nn_stub.FlopCountAnalysis = nn_stub.flop_count_table = object
    stub.nn = nn_stub
    sys.modules.setdefault("fvcore", stub)
    sys.modules.setdefault("fvcore.nn", nn_stub)
    import moce_ir
    return moce_ir


def load_adair():
    spec = importlib.util.spec_from_file_location("adair_model", os.path.join(REF, "AdaIR-main", "net", "model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def fill(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items() if not k.endswith("complexity")}
    sd = R.make_state(shapes, seed, F64)
    module.load_state_dict(sd, strict=False)
    return sd


def grads(module, inputs, seed, call):
    ins = [t.clone().requires_grad_(True) for t in inputs]
    y = call(*ins)
    y0 = y[0] if isinstance(y, tuple) else y
    cot = seeded_input(tuple(y0.shape), seed + 1000, y0.dtype)
    y0.backward(cot)
    g = {k: p.grad.detach() for k, p in module.named_parameters() if p.grad is not None}
    return y, [t.grad.detach() for t in ins], g


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    mo = load_moce()
    ad = load_adair()

    # (6) cross attention: MoCE (7x7 kv) and AdaIR (3x3)
    for tag, c, heads in (("c48h1", 48, 1), ("c96h2", 96, 2)):
        m = mo.CrossAttention(c, heads, True).double()
        fill(m, 60 + c)
        x, y = seeded_input((2, c, 16, 16), 600 + c, F64), seeded_input((2, c, 16, 16), 601 + c, F64)
        out, (dx, dy), g = grads(m, [x, y], 610, m)
        save(f"moce_cross_{tag}", y=out.detach(), dx=dx, dy=dy, **{"g_" + k: v for k, v in g.items()})
    m = ad.Chanel_Cross_Attention(48, 4, False).double()
    fill(m, 65)
    x, y = seeded_input((2, 48, 16, 16), 650, F64), seeded_input((2, 48, 16, 16), 651, F64)
    out, (dx, dy), g = grads(m, [x, y], 660, m)
    save("adair_cross_c48h4", y=out.detach(), dx=dx, dy=dy, **{"g_" + k: v for k, v in g.items()})

    # (7) routing function with injected noise; sparse dispatcher
    comp = torch.tensor([18840., 42288., 103008., 279744.])
    for k in (1, 2):
        rf = mo.RoutingFunction(48, 64, num_experts=4, k=k, complexity=comp.clone(), use_complexity_bias=True,
                                complexity_scale="max").double()
        fill(rf, 70 + k)
        x, fe = seeded_input((8, 48, 8, 8), 700, F64), seeded_input((8, 64), 701, F64)
        rf.train()
        with injected_noise(702):
            gates, idx, vals, aux = rf(x, fe)
        disp = mo.SparseDispatcher(4, gates)
        parts = disp.dispatch(x)
        comb = disp.combine([p * (e + 1) for e, p in enumerate(parts)], multiply_by_gates=True)
        save(f"moce_routing_k{k}", gates=gates.detach(), idx=idx.numpy(), vals=vals.detach(), aux=float(aux),
             part_sizes=np.array([p.shape[0] for p in parts]), combined=comb.detach())

    # EncoderBlock under MoCE names (bias=True, ffn factor 2)
    eb = mo.EncoderBlock(48, 2, 2, True, "WithBias").double()
    fill(eb, 75)
    x = seeded_input((2, 48, 16, 16), 750, F64)
    out, (dx,), g = grads(eb, [x], 760, eb)
    save("moce_encoder_c48h2", y=out.detach(), dx=dx, **{"g_" + k: v for k, v in g.items()})

    # (8) DecoderBlock, train (B=4) and eval (B=1)
    kw = dict(dim=48, num_heads=1, ffn_expansion_factor=2, bias=False, LayerNorm_type="WithBias", expert_layer=mo.FFTAttention,
              complexity_scale="max", rank=2, num_experts=4, top_k=1, depth_type="constant", rank_type="spread", stage_depth=1,
              freq_dim=64, with_complexity=True)
    # fp32, as the reference runs it: SparseDispatcher.combine accumulates in float32 (moce_ir.py:120-123), which an
    # all-fp64 module cannot consume
    db = mo.DecoderBlock(**kw)
    sd = fill(db, 80)
    db.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    x, fe = seeded_input((4, 48, 16, 16), 800), seeded_input((4, 64), 801)
    db.train()
    with injected_noise(802):
        (out, aux), (dx, dfe), g = grads(db, [x, fe], 810, db)
    save("moce_decoder_train", y=out.detach(), aux=float(aux), dx=dx, dfe=dfe, **{"g_" + k: v for k, v in g.items()},
         complexity=db.adapter.routing.complexity.numpy())
    save("moce_keys", decoder=np.array(list(db.state_dict())), encoder=np.array(list(eb.state_dict())),
         cross=np.array(list(mo.CrossAttention(48, 1, True).state_dict())),
         adair_cross=np.array(list(ad.Chanel_Cross_Attention(48, 4, False).state_dict())))
    db.eval()
    with torch.no_grad(), injected_noise(803):
        out, aux = db(x[:1], fe[:1])
    save("moce_decoder_eval", y=out.detach(), aux=float(aux))


if __name__ == "__main__":
    main()
