"""GPU parity of the fused half-block kernels (csrc/fused_gdfn.hip, csrc/fused_mdta.hip) against the fp64 CPU oracle and
against the unfused kernel chain they replace.  bf16 activations: the bound is the storage rounding (stated per test);
the fused kernels keep fewer bf16 intermediates than the chain, so they sit closer to the oracle, not further."""
import pytest
import torch

from oracle import restormer_ref as R
from oracle.fixtures import seeded_input

pytestmark = pytest.mark.gpu
DEV = "cuda"


def M():
    import image_restoration_amd as m
    return m


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _ffn_params(sd, dev):
    keys = ["ffn.project_in.weight", "ffn.project_in.bias", "ffn.dwconv.weight", "ffn.dwconv.bias",
            "ffn.project_out.weight", "ffn.project_out.bias"]
    return tuple(sd[k].to(dev).float().contiguous() if k in sd else None for k in keys)


def _oracle_half_block(y, sd, kind):
    """y + ffn(norm2(y)) in fp64 (Restormer.py:148)."""
    d = {k: v.double() for k, v in sd.items()}
    yn = R.layernorm_nchw(y.double(), d["norm2.body.weight"], d.get("norm2.body.bias"), kind)
    f = R.gdfn(yn, d["ffn.project_in.weight"], d["ffn.dwconv.weight"], d["ffn.project_out.weight"],
               d.get("ffn.project_in.bias"), d.get("ffn.dwconv.bias"), d.get("ffn.project_out.bias"))
    return y.double() + f


CASES = [
    # C, heads, ffn factor, bias, LN kind, shape                      (hidden = int(C * f))
    (48, 1, 2.66, False, "WithBias", (2, 48, 32, 64)),      # Restormer level 1: h = 127 (odd), two row tiles
    (48, 1, 2.66, True, "WithBias", (1, 48, 16, 128)),      # biases everywhere, two column tiles (left/right halo columns)
    (48, 1, 2.0, True, "BiasFree", (1, 48, 48, 192)),       # MoCE expansion 2 (h = 96), BiasFree LN, interior tiles
    (96, 2, 2.66, False, "WithBias", (2, 96, 16, 64)),      # Restormer level 2 / dec1: h = 255
    (96, 1, 2.66, True, "WithBias", (1, 96, 24, 128)),
    (96, 2, 2.0, True, "BiasFree", (1, 96, 8, 192)),
]


@pytest.mark.parametrize("c,heads,f,bias,kind,shape", CASES)
def test_gdfn_fused_fwd_vs_oracle_and_chain(c, heads, f, bias, kind, shape):
    m = M()
    from image_restoration_amd import ops
    sd = R.make_block_state(c, heads, f, bias, kind, seed=31 + c + int(bias))
    # non-trivial LayerNorm affine so that the fold into project_in is exercised
    g = torch.Generator().manual_seed(5)
    sd["norm2.body.weight"] = 1.0 + 0.3 * torch.randn(c, generator=g)
    if kind == "WithBias":
        sd["norm2.body.bias"] = 0.2 * torch.randn(c, generator=g)
    y = seeded_input(shape, 4100 + c)
    yb = y.to(DEV).to(torch.bfloat16)
    ln_w = sd["norm2.body.weight"].to(DEV).float()
    ln_b = sd["norm2.body.bias"].to(DEV).float() if kind == "WithBias" else None
    params = _ffn_params(sd, DEV)
    hidden = params[4].shape[1]
    assert ops.gdfn_fused_ok(yb, hidden)
    pack = ops.gdfn_fused_pack(yb, ln_w, ln_b, params)
    out, mean, rstd = ops.gdfn_fused_fwd(yb, pack, hidden, kind == "WithBias", want_stats=True)
    torch.cuda.synchronize()
    ref = _oracle_half_block(yb.float().cpu(), sd, kind)
    # the unfused chain on the same bf16 input
    yn, mean_c, rstd_c = ops.ln_fwd(yb, ln_w, ln_b, kind == "WithBias", want_stats=True)
    chain, _ = ops.gdfn_fwd(yn, yb, params, False)
    e_fused, e_chain = rel(out, ref), rel(chain, ref)
    assert e_fused < 2e-2, f"fused GDFN vs fp64 oracle: {e_fused:.3e} (chain {e_chain:.3e})"
    assert e_fused < 2.0 * e_chain + 4e-3, f"fused {e_fused:.3e} should not be worse than the chain {e_chain:.3e}"
    # the GDFN branch alone (out - y), so that the residual does not mask an error in it
    br_ref = ref - yb.float().cpu().double()
    e_branch = rel(out.float().cpu().double() - yb.float().cpu().double(), br_ref)
    assert e_branch < 4e-2, f"fused GDFN branch vs oracle: {e_branch:.3e}"
    assert rel(mean, mean_c) < 1e-5 and rel(rstd, rstd_c) < 1e-5


def test_gdfn_fused_rejects_uncovered_shapes():
    from image_restoration_amd import ops
    x = torch.zeros((1, 48, 16, 32), dtype=torch.bfloat16, device=DEV)      # W not a multiple of 64
    assert not ops.gdfn_fused_ok(x, 127)
    x = torch.zeros((1, 192, 16, 64), dtype=torch.bfloat16, device=DEV)     # C = 192: not covered yet
    assert not ops.gdfn_fused_ok(x, 510)
    x = torch.zeros((1, 48, 16, 64), dtype=torch.float32, device=DEV)       # fp32 stays on the exact chain
    assert not ops.gdfn_fused_ok(x, 127)


@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 32, 64)), (96, 2, (1, 96, 16, 128))])
def test_block_inference_path_uses_fused_gdfn_and_matches_training_forward(c, heads, shape):
    """TransformerBlock under no_grad (fused norm2+ffn kernel, nothing saved) vs the autograd path and the fp64 oracle;
    the packed weights follow in-place parameter updates (version check)."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=11 + c)
    blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
    blk.load_state_dict(sd)
    x = seeded_input(shape, 9000 + c).to(DEV).to(torch.bfloat16)
    ref = R.transformer_block(x.float().cpu().double(), {k: v.double() for k, v in sd.items()}, heads, "WithBias")
    y_train = blk(x)
    with torch.no_grad():
        y_inf = blk(x)
    assert getattr(blk, "_fg_pack", None) is not None, "the no_grad path did not take the fused GDFN kernel"
    assert rel(y_inf, ref) < 2e-2 and rel(y_train, ref) < 2e-2
    assert rel(y_inf, y_train.float()) < 2e-2
    # in-place weight update -> the cached pack must be rebuilt
    with torch.no_grad():
        blk.ffn.project_out.weight.mul_(0.5)
        y2 = blk(x)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["ffn.project_out.weight"] = sd2["ffn.project_out.weight"] * 0.5
    ref2 = R.transformer_block(x.float().cpu().double(), {k: v.double() for k, v in sd2.items()}, heads, "WithBias")
    assert rel(y2, ref2) < 2e-2


def test_trainer_pack_cache_follows_load_state_dict():
    """step -> load_state_dict -> forward must not run the 1x1 GEMMs on stale packed weights."""
    m = M()
    from image_restoration_amd.trainer import FlatTrainer
    torch.manual_seed(0)
    net = m.TransformerBlock(48, 1, 2.66, False, "WithBias").to(DEV)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr = FlatTrainer(net, lr=1e-2)
    x = seeded_input((1, 48, 16, 16), 77).to(DEV).to(torch.bfloat16)
    try:
        for _ in range(2):
            tr.zero_grad()
            net(x).float().abs().mean().backward()
            tr.reduce_gradients()
            tr.optimizer_step()
        y_after_steps = net(x).detach().float()
        net.load_state_dict(sd0)                        # back to the initial weights, written in place into the flat buffer
        y_loaded = net(x).detach().float()
        with torch.no_grad():                           # an in-place write that is neither a step nor a load (EMA copy-back, re-init)
            net.attn.qkv.weight.mul_(0.5)
        y_scaled = net(x).detach().float()
    finally:
        tr.close()
    fresh = m.TransformerBlock(48, 1, 2.66, False, "WithBias").to(DEV)
    fresh.load_state_dict(sd0)
    y_ref = fresh(x).detach().float()
    assert rel(y_loaded, y_ref) < 1e-6, "forward after load_state_dict used stale packed weights"
    assert rel(y_after_steps, y_ref) > 1e-4            # the steps really changed the weights
    with torch.no_grad():
        fresh.attn.qkv.weight.mul_(0.5)
    y_ref2 = fresh(x).detach().float()
    assert rel(y_ref2, y_ref) > 1e-4
    assert rel(y_scaled, y_ref2) < 1e-6, "forward after an in-place parameter write used stale packed weights"


# ------------------------------------------------------------------------------------------------ backward tail (bwd_tail.hip)
TAIL_CASES = [
    # C, M, shape of x                  (M = 3C: qkv;  M = 2 * int(2.66 C): project_in)
    (48, 144, (2, 48, 16, 64)),
    (48, 254, (3, 48, 8, 72)),        # N = 576 = 9 tiles; 254 rows: the last fragment has 14 valid rows
    (96, 288, (2, 96, 16, 32)),       # qkv at C = 96: 18 row fragments over 8 waves (two waves idle in the MFMA part)
    (96, 510, (2, 96, 24, 64)),
    (96, 510, (5, 96, 64, 64)),       # 320 tiles > 256 workgroups: the persistent loop takes a second tile
    (48, 96, (1, 48, 8, 8)),          # one tile, few rows
]


def _tail_reference(dy, x, dres, w, gamma, beta):
    """fp64 statement of what the tail computes (WithBias_LayerNorm, Restormer.py:52-64, eps 1e-5)."""
    dy, x, dres, w, gamma, beta = (t.detach().cpu().double() for t in (dy, x, dres, w, gamma, beta))
    B, Cc, H, W = x.shape
    xf, dyf = x.reshape(B, Cc, -1), dy.reshape(B, dy.shape[1], -1)
    mu = xf.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-5)
    xh = (xf - mu) * rstd
    xn = gamma[None, :, None] * xh + beta[None, :, None]
    dw = torch.einsum("bmn,bcn->mc", dyf, xn)
    dxn = torch.einsum("mc,bmn->bcn", w, dyf)
    g = dxn * gamma[None, :, None]
    dx = rstd * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True)) + dres.reshape(B, Cc, -1)
    return dx.reshape(x.shape), dw, (dxn * xh).sum((0, 2)), dxn.sum((0, 2))


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("c,mrows,shape", TAIL_CASES)
def test_bwd_tail_vs_fp64(c, mrows, shape, accumulate):
    """dW, W^T dY, LayerNorm backward and the residual add in one launch against fp64 math on the same bf16 inputs.
    Bounds: bf16 operand rounding of W and of (x - mean) rstd inside the kernel, bf16 storage of dx: 2e-2 of the tensor's
    largest magnitude for dx, 1e-2 for the three parameter gradients (fp32 accumulation over all pixels)."""
    from image_restoration_amd import ops
    B, Cc, H, W = shape
    assert ops.bwd_tail_ok(mrows, c, H * W, torch.bfloat16)
    x = (seeded_input(shape, 100 + mrows) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    dy = seeded_input((B, mrows, H, W), 200 + mrows).to(DEV).to(torch.bfloat16)
    dres = seeded_input(shape, 300 + mrows).to(DEV).to(torch.bfloat16)
    w = (seeded_input((mrows, c), 400 + mrows) * 0.2).to(DEV).float().contiguous()
    gamma = (1.0 + 0.3 * seeded_input((c,), 500 + c)).to(DEV).float()
    beta = (0.2 * seeded_input((c,), 600 + c)).to(DEV).float()
    _, mean, rstd = ops.ln_fwd(x, gamma, beta, True, want_stats=True)
    init = 0.5 if accumulate else float("nan")                       # overwrite mode must not read the old contents
    dw = torch.full((mrows, c), init, device=DEV)
    dgamma = torch.full((c,), init, device=DEV)
    dbeta = torch.full((c,), init, device=DEV)
    dx = ops.bwd_tail(dy, x, dres, mean, rstd, w, gamma, beta, dw, dgamma, dbeta, accumulate)
    torch.cuda.synchronize()
    rdx, rdw, rdg, rdb = _tail_reference(dy, x, dres, w, gamma, beta)
    off = 0.5 if accumulate else 0.0
    assert rel(dx, rdx) < 2e-2, rel(dx, rdx)
    assert rel(dw - off, rdw) < 1e-2, rel(dw - off, rdw)
    assert rel(dgamma - off, rdg) < 1e-2, rel(dgamma - off, rdg)
    assert rel(dbeta - off, rdb) < 1e-2, rel(dbeta - off, rdb)


@pytest.mark.parametrize("seed", range(8))
def test_bwd_tail_random_shapes(seed):
    """Random covered shapes (rows not a multiple of 16, odd tile counts, 1-5 images) against the fp64 statement."""
    import numpy as np
    from image_restoration_amd import ops
    rng = np.random.default_rng(1000 + seed)
    c = int(rng.choice([48, 96]))
    mrows = int(rng.integers(17, 256 if c == 48 else 512))
    B, tiles = int(rng.integers(1, 6)), int(rng.integers(1, 40))
    H, W = 8, 8 * tiles                                          # H*W = 64 * tiles
    assert ops.bwd_tail_ok(mrows, c, H * W, torch.bfloat16)
    x = (seeded_input((B, c, H, W), 2000 + seed) * 2.0 - 0.5).to(DEV).to(torch.bfloat16)
    dy = seeded_input((B, mrows, H, W), 2100 + seed).to(DEV).to(torch.bfloat16)
    dres = seeded_input((B, c, H, W), 2200 + seed).to(DEV).to(torch.bfloat16)
    w = (seeded_input((mrows, c), 2300 + seed) * 0.3).to(DEV).float().contiguous()
    gamma = (1.0 + 0.3 * seeded_input((c,), 2400 + seed)).to(DEV).float()
    beta = (0.2 * seeded_input((c,), 2500 + seed)).to(DEV).float()
    _, mean, rstd = ops.ln_fwd(x, gamma, beta, True, want_stats=True)
    dw, dgamma, dbeta = (torch.full(s_, float("nan"), device=DEV) for s_ in ((mrows, c), (c,), (c,)))
    dx = ops.bwd_tail(dy, x, dres, mean, rstd, w, gamma, beta, dw, dgamma, dbeta, False)
    rdx, rdw, rdg, rdb = _tail_reference(dy, x, dres, w, gamma, beta)
    assert rel(dx, rdx) < 2e-2 and rel(dw, rdw) < 1e-2 and rel(dgamma, rdg) < 1e-2 and rel(dbeta, rdb) < 1e-2, (c, mrows, B, tiles)


def test_bwd_tail_rejects_uncovered_shapes():
    from image_restoration_amd import ops
    assert not ops.bwd_tail_ok(576, 192, 4096, torch.bfloat16)       # C = 192: unfused chain
    assert not ops.bwd_tail_ok(288, 96, 4096, torch.float32)         # fp32 stays on the exact chain
    assert not ops.bwd_tail_ok(288, 96, 4000, torch.bfloat16)        # pixel count not a multiple of the 64-pixel tile
    assert not ops.bwd_tail_ok(600, 96, 4096, torch.bfloat16)        # more rows than eight waves hold


@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 16, 64)), (96, 2, (2, 96, 16, 32))])
def test_block_backward_with_and_without_tail(c, heads, shape, monkeypatch):
    """TransformerBlock forward + backward through the tail kernels and through the unfused chain (MI_NO_BWD_TAIL=1): both
    within the bf16 bound of the fp64 oracle, for dx and for every parameter gradient (at C = 96 the GDFN half takes the
    4-fragment form of the kernel, 510 rows)."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=31 + c)
    x0 = seeded_input(shape, 7000 + c)
    cot = seeded_input(shape, 7001 + c)
    xr = x0.double().requires_grad_(True)
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    R.transformer_block(xr, ps, heads, "WithBias").backward(cot.double())
    got = {}
    for mode in ("tail", "chain"):
        if mode == "chain":
            monkeypatch.setenv("MI_NO_BWD_TAIL", "1")
        blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
        blk.load_state_dict(sd)
        x = x0.to(DEV).to(torch.bfloat16).requires_grad_(True)
        blk(x).backward(cot.to(DEV).to(torch.bfloat16))
        got[mode] = (x.grad, {k: p.grad for k, p in blk.named_parameters()})
        assert rel(x.grad, xr.grad) < 3e-2, (mode, rel(x.grad, xr.grad))
        for k, gk in got[mode][1].items():
            assert rel(gk, ps[k].grad) < 3e-2, (mode, k, rel(gk, ps[k].grad))
    assert rel(got["tail"][0], got["chain"][0].float()) < 3e-2


# ------------------------------------------------------------------------------------------------ LayerNorm inside the first 1x1 conv
@pytest.mark.parametrize("kind", ["WithBias", "BiasFree"])
@pytest.mark.parametrize("c,heads,f,shape", [(48, 1, 2.66, (2, 48, 16, 64)), (96, 2, 2.66, (2, 96, 8, 64)), (64, 2, 2.0, (1, 64, 16, 32)),
                                             (128, 4, 2.66, (2, 128, 8, 72)), (112, 2, 2.5, (1, 112, 16, 64))])   # 96 < K <= 128: the W-streamed form
def test_ln_head_inside_first_gemm(c, heads, f, shape, kind):
    """mi_mdta_fwd_ln / mi_gdfn_fwd_ln (LayerNorm applied as the GEMM loads its tile) against ln_fwd followed by the plain entry
    points: same arithmetic, so the two agree to a few bf16 ulps (5e-3 of the largest magnitude; the summation order of the
    statistics differs), and the statistics agree to 1e-5."""
    from image_restoration_amd import ops
    sd = R.make_block_state(c, heads, f, False, kind, seed=51 + c)
    dev = lambda k: sd[k].to(DEV).float().contiguous() if k in sd else None
    att = (dev("attn.temperature"), dev("attn.qkv.weight"), dev("attn.qkv.bias"), dev("attn.qkv_dwconv.weight"),
           dev("attn.qkv_dwconv.bias"), dev("attn.project_out.weight"), dev("attn.project_out.bias"))
    ffn = _ffn_params(sd, DEV)
    x = (seeded_input(shape, 8100 + c) * 1.7 + 0.4).to(DEV).to(torch.bfloat16)
    wb = kind == "WithBias"
    n1 = (dev("norm1.body.weight"), dev("norm1.body.bias"))
    n2 = (dev("norm2.body.weight"), dev("norm2.body.bias"))
    assert ops.mdta_fwd_ln_ok(x, heads, 3) and ops.gdfn_fwd_ln_ok(x, ffn[4].shape[1], 3)
    xn, mean, rstd = ops.ln_fwd(x, n1[0], n1[1], wb, want_stats=True)
    ref, _ = ops.mdta_fwd(xn, x, att, heads, False)
    got, _, m2, r2 = ops.mdta_fwd(x, x, att, heads, False, ln=(n1[0], n1[1], True))
    assert rel(got, ref.float()) < 5e-3, rel(got, ref.float())
    assert rel(m2, mean) < 1e-5 and rel(r2, rstd) < 1e-5
    yn, mean, rstd = ops.ln_fwd(x, n2[0], n2[1], wb, want_stats=True)
    ref, _ = ops.gdfn_fwd(yn, x, ffn, False)
    got, _, m2, r2 = ops.gdfn_fwd(x, x, ffn, False, ln=(n2[0], n2[1], True))
    assert rel(got, ref.float()) < 5e-3, rel(got, ref.float())
    assert rel(m2, mean) < 1e-5 and rel(r2, rstd) < 1e-5


def test_ln_head_rejects_uncovered_shapes():
    from image_restoration_amd import ops
    assert not ops.mdta_fwd_ln_ok(torch.zeros((1, 192, 16, 64), dtype=torch.bfloat16, device=DEV), 4, 3)   # K > 128
    assert not ops.mdta_fwd_ln_ok(torch.zeros((1, 48, 16, 64), dtype=torch.float32, device=DEV), 1, 3)     # fp32
    assert not ops.mdta_fwd_ln_ok(torch.zeros((1, 48, 10, 10), dtype=torch.bfloat16, device=DEV), 1, 3)    # ragged plane


# ------------------------------------------------------------------------------------------------ round-2 A/B switches, bf16 step path
@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 16, 64)), (192, 4, (2, 192, 16, 64)), (384, 8, (3, 384, 16, 64))])
def test_round2_switches_leave_the_block_unchanged(c, heads, shape, monkeypatch):
    """The bf16 training path of a TransformerBlock under the switches that select this round's alternative kernels: same output,
    input gradient and parameter gradients as the default.  Exact where only the launch structure differs (q / k gradients as one
    GEMM or two, per-image weights packed or staged in the GEMM); to 2e-2 of the largest value where a summation order changes
    (weight-gradient Grams folded over the batch or not, the W-streamed GEMM form or the streaming one)."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=91 + c)
    x0, cot = seeded_input(shape, 7100 + c), seeded_input(shape, 7101 + c)

    def run_block():
        blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
        blk.load_state_dict(sd)
        x = x0.to(DEV).to(torch.bfloat16).requires_grad_(True)
        y = blk(x)
        y.backward(cot.to(DEV).to(torch.bfloat16))
        return [y.detach().float(), x.grad.float()] + [p.grad.float() for _, p in sorted(blk.named_parameters())]

    base = run_block()
    for env, exact in (("MI_ATTN_DQK_SPLIT=1", True), ("MI_PW_DIRECT=1", True), ("MI_GRAM_FOLD=0", False), ("MI_GRAM_FOLD=2", False),
                       ("MI_PW_XWIDE=0", False)):
        name, _, val = env.partition("=")
        monkeypatch.setenv(name, val)
        got = run_block()
        monkeypatch.delenv(name)
        for i, (a, b) in enumerate(zip(got, base)):
            if exact:
                assert torch.equal(a, b), (env, i, float((a - b).abs().max()))
            else:
                assert rel(a, b) < 2e-2, (env, i, rel(a, b))


@pytest.mark.parametrize("c,heads,hw", [(96, 1, (256, 256)), (192, 4, (64, 64)), (384, 8, (32, 32))])
def test_full_size_batch_is_consistent_with_a_small_one(c, heads, hw):
    """BASELINE's full per-GPU batch (32) at the step's real planes, through a property that needs no oracle at that size: a batch
    of 16 copies of image A and 16 of image B must give, per image, what a batch [A, B] gives (every kernel is per-image in the
    forward and in the input gradient; the launch plans - pixel splits, tiles per wave, M splits, folded Grams - change with the
    batch, so sums are taken in another order: equal to bf16 rounding, 1e-2 / 2e-2 of the largest value, and EXACTLY equal between
    copies inside the big batch), and parameter gradients 16 times those of the small batch (2e-2).  The small batch itself is held to the oracle elsewhere."""
    m = M()
    H, W = hw
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=131 + c)
    xs = seeded_input((2, c, H, W), 7300 + c)
    cs = seeded_input((2, c, H, W), 7301 + c)

    def run_block(x0, cot):
        blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
        blk.load_state_dict(sd)
        x = x0.to(DEV).to(torch.bfloat16).requires_grad_(True)
        y = blk(x)
        y.backward(cot.to(DEV).to(torch.bfloat16))
        return y.detach(), x.grad, {k: p.grad.float() for k, p in blk.named_parameters()}

    ys, dxs, gs = run_block(xs, cs)
    idx = torch.tensor([0] * 16 + [1] * 16)
    yb, dxb, gb = run_block(xs[idx], cs[idx])
    for i in (0, 7, 15, 16, 31):
        assert rel(yb[i], ys[idx[i]].float()) < 1e-2, ("forward", i, rel(yb[i], ys[idx[i]].float()))
        assert rel(dxb[i], dxs[idx[i]].float()) < 2e-2, ("input gradient", i, rel(dxb[i], dxs[idx[i]].float()))
    assert torch.equal(yb[0], yb[15]) and torch.equal(yb[16], yb[31]) and torch.equal(dxb[0], dxb[15])   # copies inside one batch: exact
    for k in gs:
        assert rel(gb[k], 16.0 * gs[k]) < 2e-2, (k, rel(gb[k], 16.0 * gs[k]))


# ------------------------------------------------------------------------------------------------ fused MDTA, pass A (csrc/fused_mdta.hip)
def _oracle_attn_half(x, sd, heads, kind):
    """x + attn(norm1(x)) in fp64 (Restormer.py:147)."""
    d = {k: v.double() for k, v in sd.items()}
    xn = R.layernorm_nchw(x.double(), d["norm1.body.weight"], d.get("norm1.body.bias"), kind)
    return x.double() + R.mdta(xn, d["attn.temperature"], d["attn.qkv.weight"], d["attn.qkv_dwconv.weight"],
                               d["attn.project_out.weight"], heads, d.get("attn.qkv.bias"), d.get("attn.qkv_dwconv.bias"),
                               d.get("attn.project_out.bias"))


FM_CASES = [
    # C, heads, bias, LN kind, shape
    (48, 1, False, "WithBias", (2, 48, 16, 64)),
    (48, 1, True, "BiasFree", (1, 48, 8, 128)),
    (96, 2, False, "WithBias", (2, 96, 24, 64)),
    (96, 1, False, "WithBias", (2, 96, 16, 64)),
    (96, 1, True, "WithBias", (3, 96, 40, 128)),       # 3 images x 20 tiles over 32 splits: uneven tile ranges, workgroups with one tile
    (96, 2, True, "BiasFree", (1, 96, 64, 64)),
]


@pytest.mark.parametrize("c,heads,bias,kind,shape", FM_CASES)
def test_mdta_fused_pass_a_vs_oracle_and_chain(c, heads, bias, kind, shape):
    """x + attn(norm1(x)) with LN -> qkv -> dw3x3 -> q k^T in ONE launch (q, k, qkv0 never written) against the fp64 oracle and
    against the unfused kernel chain it replaces; the LayerNorm statistics it can emit against the LayerNorm kernel's."""
    m = M()
    from image_restoration_amd import ops
    sd = R.make_block_state(c, heads, 2.66, bias, kind, seed=130 + c + heads)
    x = seeded_input(shape, 1300 + c).to(DEV).to(torch.bfloat16)
    ref = _oracle_attn_half(x.float().cpu(), sd, heads, kind)
    ln = (sd["norm1.body.weight"].to(DEV), sd["norm1.body.bias"].to(DEV) if "norm1.body.bias" in sd else None)
    keys = ["attn.temperature", "attn.qkv.weight", "attn.qkv.bias", "attn.qkv_dwconv.weight", "attn.qkv_dwconv.bias",
            "attn.project_out.weight", "attn.project_out.bias"]
    att = tuple(sd[k].to(DEV).float().contiguous() if k in sd else None for k in keys)
    assert ops.mdta_fused_ok(x, heads, 3)
    pack = ops.mdta_fused_pack(x, heads, ln[0], ln[1], att)
    y, mean, rstd = ops.mdta_fused_fwd(x, pack, att, heads, kind == "WithBias", x, want_stats=True)
    xn, mean_r, rstd_r = ops.ln_fwd(x, ln[0], ln[1], kind == "WithBias", want_stats=True)
    chain, _ = ops.mdta_fwd(xn, x, att, heads, False)
    e_or, e_ch = rel(y, ref), rel(chain, ref)
    assert e_or < 2e-2, (e_or, e_ch)
    assert e_or < 1.5 * e_ch + 4e-3, (e_or, e_ch)            # no further from the oracle than the chain (fewer bf16 intermediates)
    assert rel(mean, mean_r) < 1e-5 and rel(rstd, rstd_r) < 1e-4
    y2, _, _ = ops.mdta_fused_fwd(x, pack, att, heads, kind == "WithBias", x)
    assert torch.equal(y, y2)                                # fixed-order partial sums: bit-reproducible


def test_block_infer_takes_the_fused_mdta_kernel(monkeypatch):
    """TransformerBlock.forward under no_grad: both halves are one-launch kernels where the shapes allow; output vs the oracle.
    (The fused MDTA pass is taken by default only where it fills the chip - mi_mdta_fused_pays; this small case forces it.)"""
    m = M()
    from image_restoration_amd import ops
    small = torch.zeros((2, 96, 32, 64), dtype=torch.bfloat16, device=DEV)
    big = torch.zeros((8, 96, 256, 256), dtype=torch.bfloat16, device=DEV)
    assert ops.mdta_fused_ok(small, 1) and not ops.mdta_fused_pays(small, 1) and ops.mdta_fused_pays(big, 1)
    assert not ops.mdta_fused_pays(torch.zeros((8, 96, 128, 128), dtype=torch.bfloat16, device=DEV), 2)   # 128 workgroups: the chain wins
    del big
    monkeypatch.setenv("MI_FUSED_MDTA_ALWAYS", "1")
    m.reload_env()
    c, heads, shape = 96, 1, (2, 96, 32, 64)
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=141)
    blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
    blk.load_state_dict(sd)
    x = seeded_input(shape, 1410).to(DEV).to(torch.bfloat16)
    ref = R.transformer_block(x.float().cpu().double(), {k: v.double() for k, v in sd.items()}, heads, "WithBias")
    with torch.no_grad():
        y = blk(x)
    assert getattr(blk, "_fm_pack", None) is not None and getattr(blk, "_fg_pack", None) is not None
    assert rel(y, ref) < 2e-2, rel(y, ref)
    with torch.no_grad():                                   # a weight written in place: the pack follows
        blk.attn.qkv.weight.mul_(0.5)
        y2 = blk(x)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["attn.qkv.weight"] = sd2["attn.qkv.weight"] * 0.5
    ref2 = R.transformer_block(x.float().cpu().double(), {k: v.double() for k, v in sd2.items()}, heads, "WithBias")
    assert rel(y2, ref2) < 2e-2


# ------------------------------------------------------------------------------------------------ fused GDFN forward on the training path
@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 32, 64)), (48, 1, (1, 48, 16, 128)), (96, 2, (2, 96, 16, 128)),
                                           (96, 1, (1, 96, 24, 64))])
def test_gdfn_fused_training_forward_saves_what_the_chain_saves(c, heads, shape):
    """mi_gdfn_fused_fwd_train: out, the LayerNorm statistics and the saved blob (project_in output h0, gate output g) against
    the chain mi_gdfn_fwd_ln on the same input.  Both round h0 and g to bf16 once from fp32 accumulators, so the blobs agree to
    a bf16 ulp of the largest entry; out against the fp64 oracle within the bf16 bound."""
    from image_restoration_amd import ops
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=61 + c)
    g = torch.Generator().manual_seed(7)
    sd["norm2.body.weight"] = 1.0 + 0.3 * torch.randn(c, generator=g)
    sd["norm2.body.bias"] = 0.2 * torch.randn(c, generator=g)
    yb = seeded_input(shape, 5100 + c).to(DEV).to(torch.bfloat16)
    ln_w, ln_b = sd["norm2.body.weight"].to(DEV).float(), sd["norm2.body.bias"].to(DEV).float()
    params = _ffn_params(sd, DEV)
    hidden = params[4].shape[1]
    assert ops.gdfn_fused_train_ok(yb, hidden)
    out, saved, mean, rstd = ops.gdfn_fused_fwd_train(yb, ops.gdfn_fused_pack(yb, ln_w, ln_b, params), hidden, True)
    out_c, saved_c, mean_c, rstd_c = ops.gdfn_fwd(yb, yb, params, True, ln=(ln_w, ln_b, True))
    torch.cuda.synchronize()
    assert saved.numel() == saved_c.numel()
    B, _, H, W = shape
    n_h0 = B * 2 * hidden * H * W
    h0, h0_c = saved[: 2 * n_h0].view(torch.bfloat16).float(), saved_c[: 2 * n_h0].view(torch.bfloat16).float()
    off_g = (2 * n_h0 + 255) // 256 * 256
    gg, gg_c = (t[off_g: off_g + n_h0].view(torch.bfloat16).float() for t in (saved, saved_c))
    assert rel(h0, h0_c) < 1.2e-2, rel(h0, h0_c)          # LN affine folded into W (fused) vs applied to x (chain): bf16 operand rounding
    assert rel(gg, gg_c) < 2e-2, rel(gg, gg_c)
    ref = _oracle_half_block(yb.float().cpu(), sd, "WithBias")
    assert rel(out, ref) < 2e-2 and rel(out_c, ref) < 2e-2
    assert rel(mean, mean_c) < 1e-5 and rel(rstd, rstd_c) < 1e-5
    # h0 against the oracle's project_in(LN(y)) directly
    d = {k: v.double() for k, v in sd.items()}
    yn = R.layernorm_nchw(yb.float().cpu().double(), d["norm2.body.weight"], d["norm2.body.bias"], "WithBias")
    h0_ref = torch.nn.functional.conv2d(yn, d["ffn.project_in.weight"])
    assert rel(h0.view(B, 2 * hidden, H, W), h0_ref) < 1.2e-2


@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 32, 64)), (96, 2, (2, 96, 16, 64))])
def test_block_training_step_with_fused_gdfn_forward(c, heads, shape, monkeypatch):
    """TransformerBlock forward + backward with the second half-block's forward in one launch (MI_FUSED_TRAIN=1) and on the
    chain (default): dx and every parameter gradient within the bf16 bound of the fp64 oracle in both modes."""
    m = M()
    from image_restoration_amd import restormer
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=71 + c)
    x0, cot = seeded_input(shape, 7100 + c), seeded_input(shape, 7101 + c)
    xr = x0.double().requires_grad_(True)
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref_out = R.transformer_block(xr, ps, heads, "WithBias")
    ref_out.backward(cot.double())
    outs = {}
    for mode in ("fused", "chain"):
        if mode == "fused":
            monkeypatch.setenv("MI_FUSED_TRAIN", "1")
        else:
            monkeypatch.delenv("MI_FUSED_TRAIN")
        m.reload_env()
        blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias").to(DEV)
        blk.load_state_dict(sd)
        x = x0.to(DEV).to(torch.bfloat16).requires_grad_(True)
        params = blk.norm1._params() + blk.attn._params() + blk.norm2._params() + blk.ffn._params()
        assert restormer._block_plan(x, heads, params, True)["fused_f"] == (mode == "fused")
        y = blk(x)
        y.backward(cot.to(DEV).to(torch.bfloat16))
        outs[mode] = y
        assert rel(y, ref_out) < 2e-2, (mode, rel(y, ref_out))
        assert rel(x.grad, xr.grad) < 3e-2, (mode, rel(x.grad, xr.grad))
        for k, p in blk.named_parameters():
            assert rel(p.grad, ps[k].grad) < 3e-2, (mode, k, rel(p.grad, ps[k].grad))
    assert rel(outs["fused"], outs["chain"].float()) < 2e-2
