"""Pins the CPU oracle (oracle/restormer_ref.py) against golden vectors captured from the
imported reference (tools/capture_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input

F64 = torch.float64
TOL = 1e-9  # fp64 oracle vs fp64 reference; fixtures store fp32 subsets -> 1e-6 on those


def _grads(fn, x, params, seed):
    x = x.clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    y = fn(x, ps)
    y.backward(seeded_input(tuple(y.shape), seed + 1000, y.dtype))
    return y.detach(), x.grad, {k: v.grad for k, v in ps.items()}


def _cmp(gold, y, dx, g, rtol=2e-6):
    check("y", y, gold, rtol)
    check("dx", dx, gold, rtol)
    for k, v in g.items():
        check("g_" + k, v, gold, rtol)


@pytest.mark.parametrize("kind", ["WithBias", "BiasFree"])
@pytest.mark.parametrize("c", [16, 48, 96])
def test_layernorm(kind, c):
    rng = np.random.default_rng(10 + c)
    sd = {"body.weight": R.seeded_tensor(rng, (c,), "ln_w", F64)}
    if kind == "WithBias":
        sd["body.bias"] = R.seeded_tensor(rng, (c,), "ln_b", F64)
    x = seeded_input((1, c, 8, 8), 100 + c, F64) * 1.5 + 0.3
    y, dx, g = _grads(lambda t, p: R.layernorm_nchw(t, p["body.weight"], p.get("body.bias"), kind), x, sd, 100 + c)
    _cmp(load(f"ln_{kind}_{c}"), y, dx, g)


CASES = [("c48h1", 48, 1, (2, 48, 16, 16), False), ("c48h1_bias", 48, 1, (2, 48, 16, 16), True),
         ("c16h1", 16, 1, (2, 16, 16, 16), False), ("c96h2", 96, 2, (2, 96, 16, 16), False),
         ("c96h1", 96, 1, (2, 96, 16, 16), False), ("c48h1_64", 48, 1, (1, 48, 64, 64), False)]


def _ffn(t, p):
    return R.gdfn(t, p["project_in.weight"], p["dwconv.weight"], p["project_out.weight"],
                  p.get("project_in.bias"), p.get("dwconv.bias"), p.get("project_out.bias"))


def _attn(heads):
    def f(t, p):
        return R.mdta(t, p["temperature"], p["qkv.weight"], p["qkv_dwconv.weight"], p["project_out.weight"], heads,
                      p.get("qkv.bias"), p.get("qkv_dwconv.bias"), p.get("project_out.bias"))
    return f


@pytest.mark.parametrize("tag,c,heads,shape,bias", CASES)
def test_ffn_attn_block(tag, c, heads, shape, bias):
    sd = R.make_block_state(c, heads, 2.66, bias, "WithBias", seed=7 + c + heads, dtype=F64)
    x = seeded_input(shape, 200 + c + heads, F64)
    _cmp(load("ffn_" + tag), *_grads(_ffn, x, R.sub_state(sd, "ffn."), 300))
    _cmp(load("attn_" + tag), *_grads(_attn(heads), x, R.sub_state(sd, "attn."), 400))
    _cmp(load("block_" + tag), *_grads(lambda t, p: R.transformer_block(t, p, heads, "WithBias"), x, sd, 500))


def test_block_biasfree():
    sd = R.make_block_state(48, 1, 2.66, False, "BiasFree", seed=77, dtype=F64)
    x = seeded_input((2, 48, 16, 16), 277, F64)
    _cmp(load("block_c48h1_biasfree"), *_grads(lambda t, p: R.transformer_block(t, p, 1, "BiasFree"), x, sd, 500))


def test_restormer_tiny_config_c1():
    """BASELINE config 1: Restormer-tiny, sigma=25 denoise, one 128x128 patch, CPU forward."""
    gold = load("restormer_tiny_128")
    cfg = R.RESTORMER_TINY
    sd = R.make_restormer_state(cfg, seed=1)
    clean = torch.from_numpy(np.random.default_rng(1234).random((1, 3, 128, 128))).to(torch.float32)
    noisy = R.degrade_sigma(clean, 25.0, seed=4321)
    assert abs(R.psnr(noisy, clean) - float(gold["psnr_in"])) < 1e-9
    with torch.no_grad():
        y32 = R.restormer_forward(noisy, sd, cfg)
    check("y32", y32[:, :, ::4, ::4], gold, 1e-4)      # fp32 oracle vs fp32 reference
    check("y64", y32[:, :, ::4, ::4], gold, 1e-3)      # the 1e-3 parity bar vs the fp64 run
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    out = R.restormer_forward(noisy.double(), sd64, cfg)
    check("y64", out[:, :, ::4, ::4], gold, 2e-6)
    assert abs(R.psnr(out, clean) - float(gold["psnr_out"])) < 1e-6
    loss = (out - clean.double()).abs().mean()
    assert abs(loss.item() - float(gold["loss"])) < 1e-9
    loss.backward()
    keys = [str(k) for k in gold["grad_norm_keys"]]
    ref = np.asarray(gold["grad_norms"])
    got = np.array([float(sd64[k].grad.norm()) for k in keys])
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-12)


def test_base_keys_and_param_count():
    gold = load("restormer_base_keys")
    shapes = R.restormer_param_shapes(R.RESTORMER_BASE)
    assert [str(k) for k in gold["keys"]] == list(shapes)
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == int(gold["n_params"]) == 26126644
