"""GPU parity of the kernel archetypes against the CPU oracle / plain torch fp64 restatements.
fp32 activations: exact-fp32 MFMA path, tolerance 2e-5 relative to max|ref| (GEMM reductions up to K~2000);
bf16 activations: bf16 storage, fp32 accumulate, tolerance 2e-2.  Integer-valued data must be EXACT
(catches any MFMA fragment / transpose-read mapping error)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import restormer_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2.5e-2}


def ops():
    from image_restoration_amd import ops as o
    return o


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def ints(shape, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


# --------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("wb", [True, False])
@pytest.mark.parametrize("shape", [(2, 48, 16, 16), (1, 16, 5, 7), (2, 96, 8, 32), (1, 384, 4, 8), (1, 192, 3, 3)])
def test_layernorm_fwd_bwd(dtype, wb, shape):
    o = ops()
    B, C, H, W = shape
    x = (rnd(shape, 1) * 1.5 + 0.3).to(dtype)
    w = 1.0 + 0.2 * rnd((C,), 2)
    b = 0.1 * rnd((C,), 3) if wb else None
    dy = rnd(shape, 4).to(dtype)
    dres = rnd(shape, 5).to(dtype)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if wb else None
    yr = R.layernorm_nchw(xr, wr, br, "WithBias" if wb else "BiasFree")
    yr.backward(dy.double())
    y, mean, rstd = o.ln_fwd(x.to(DEV), w.to(DEV), b.to(DEV) if wb else None, wb)
    assert rel(y, yr) < TOL[dtype]
    dw = torch.zeros(C, device=DEV)
    db = torch.zeros(C, device=DEV) if wb else None
    dx = o.ln_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dres.to(DEV), wb, dw, db, False)
    assert rel(dx, xr.grad + dres.double()) < TOL[dtype]
    assert rel(dw, wr.grad) < TOL[dtype]
    if wb:
        assert rel(db, br.grad) < TOL[dtype]
    # accumulate mode adds on top
    dw2 = dw.clone()
    o.ln_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, None, wb, dw2, db.clone() if wb else None, True)
    assert rel(dw2, 2 * wr.grad) < TOL[dtype]


# --------------------------------------------------------------------------- depthwise conv
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("ks", [3, 7])
@pytest.mark.parametrize("shape", [(2, 6, 16, 64), (1, 5, 33, 70), (2, 8, 32, 32), (1, 4, 9, 13), (1, 3, 16, 16)])
def test_dwconv_fwd_bwd(dtype, ks, shape):
    o = ops()
    B, C, H, W = shape
    x = rnd(shape, 11).to(dtype)
    w = rnd((C, 1, ks, ks), 12) / ks
    bias = 0.1 * rnd((C,), 13)
    dy = rnd(shape, 14).to(dtype)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2, groups=C)
    yr.backward(dy.double())
    y = o.dwconv_fwd(x.to(DEV), w.to(DEV), bias.to(DEV))
    assert rel(y, yr) < TOL[dtype]
    dx, dw, db = o.dwconv_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), True)
    assert rel(dx, xr.grad) < TOL[dtype]
    assert rel(dw, wr.grad) < TOL[dtype]
    assert rel(db, br.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 10, 16, 64), (1, 254, 16, 16), (1, 6, 7, 9)])
def test_dwconv_gate_fwd_bwd(dtype, shape):
    o = ops()
    B, C2, H, W = shape
    h = C2 // 2
    x = rnd(shape, 21).to(dtype)
    w = rnd((C2, 1, 3, 3), 22) / 3
    dg = rnd((B, h, H, W), 24).to(dtype)
    y, g = o.dwconv_gate_fwd(x.to(DEV), w.to(DEV), None)
    yr = F.conv2d(x.double(), w.double(), None, padding=1, groups=C2)
    assert rel(y, yr) < TOL[dtype]
    # the gate is evaluated on y as stored; backward oracle starts from the stored y as well
    ys = y.detach().cpu().double().requires_grad_(True)
    gr = F.gelu(ys[:, :h]) * ys[:, h:]
    assert rel(g, gr) < TOL[dtype]
    gr.backward(dg.double())
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1, groups=C2).backward(ys.grad)
    dx, dw, db = o.dwconv_gate_bwd(dg.to(DEV), y, x.to(DEV), w.to(DEV), True)
    assert rel(dx, xr.grad) < TOL[dtype]
    assert rel(dw, wr.grad) < TOL[dtype]
    assert rel(db, ys.grad.sum(dim=(0, 2, 3))) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("shape", [(2, 10, 16, 64), (1, 254, 16, 16), (2, 6, 40, 256), (1, 4, 9, 32), (3, 2, 5, 128)])
def test_dwconv_gate_bwd_recompute(dtype, bias, shape):
    """Gate backward with y recomputed from the conv input (no stored y): dx, dW, db against autograd through
    conv -> chunk -> gelu(y1) * y2 in fp64; band edges (H not a multiple of the band), several bands, all row widths."""
    o = ops()
    B, C2, H, W = shape
    h = C2 // 2
    assert o.dwconv_gate_recompute_ok(H, W, 3)
    x = rnd(shape, 41).to(dtype)
    w = rnd((C2, 1, 3, 3), 42) / 3
    bv = 0.1 * rnd((C2,), 43) if bias else None
    dg = rnd((B, h, H, W), 44).to(dtype)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = bv.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=1, groups=C2)
    (F.gelu(yr[:, :h]) * yr[:, h:]).backward(dg.double())
    dx, dw, db = o.dwconv_gate_bwd_recompute(dg.to(DEV), x.to(DEV), w.to(DEV), bv.to(DEV) if bias else None)
    assert rel(dx, xr.grad) < TOL[dtype]
    assert rel(dw, wr.grad) < TOL[dtype]
    if bias:
        assert rel(db, br.grad) < TOL[dtype]
    assert not o.dwconv_gate_recompute_ok(7, 9, 3) and not o.dwconv_gate_recompute_ok(16, 16, 7)


# --------------------------------------------------------------------------- pointwise GEMM
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("M,K", [(16, 16), (48, 48), (144, 48), (254, 48), (48, 127), (288, 96), (130, 70), (1152, 384)])
def test_conv1x1_exact_on_integers(dtype, transposed, M, K):
    """small-integer operands: every product and partial sum is exactly representable -> results must be EXACT.
    The weight is asymmetric and non-square so a swapped row/col or a permuted k would show."""
    o = ops()
    B, H, W = 2, 8, 16
    x = ints((B, K, H, W), 31).to(dtype)
    wmat = ints((M, K), 32, -2, 3)
    res = ints((B, M, H, W), 33).to(dtype)
    bias = ints((M,), 34)
    ref = torch.einsum("mk,bkhw->bmhw", wmat, x.float()) + bias.view(1, -1, 1, 1) + res.float()
    warg = wmat.t().contiguous() if transposed else wmat
    y = o.conv1x1(x.to(DEV), warg.to(DEV), bias.to(DEV), res.to(DEV), transposed)
    ref = ref.to(dtype).float()  # the fp32 accumulator is exact; only the final store rounds (bf16: |v| > 256)
    assert torch.equal(y.float().cpu(), ref), f"max diff {(y.float().cpu() - ref).abs().max()}"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1, 48, 5, 7), (2, 96, 3, 10), (1, 20, 16, 16)])
def test_conv1x1_ragged_and_two_panels(dtype, shape):
    """N not a multiple of the vector width / tile, and the concat-free two-panel form."""
    o = ops()
    B, K, H, W = shape
    M, K2 = 37, 24
    x1 = rnd(shape, 41).to(dtype)
    x2 = rnd((B, K2, H, W), 42).to(dtype)
    w = rnd((M, K + K2), 43) / math.sqrt(K + K2)
    ref = torch.einsum("mk,bkhw->bmhw", w.double(), torch.cat([x1, x2], 1).double())
    y = o.conv1x1(x1.to(DEV), w.to(DEV), None, None, False, x2.to(DEV))
    assert rel(y, ref) < TOL[dtype]
    y1 = o.conv1x1(x1.to(DEV), w[:, :K].contiguous().to(DEV))
    assert rel(y1, torch.einsum("mk,bkhw->bmhw", w[:, :K].double(), x1.double())) < TOL[dtype]


# --------------------------------------------------------------------------- Gram
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("ma,mb,groups,hw", [(48, 48, 1, (16, 16)), (48, 48, 2, (8, 40)), (96, 96, 1, (32, 32)),
                                             (16, 16, 8, (4, 4)), (254, 48, 1, (16, 24)), (30, 70, 1, (5, 7))])
def test_gram_exact_on_integers_and_sumsq(dtype, ma, mb, groups, hw):
    o = ops()
    B = 2
    a = ints((B, groups * ma, *hw), 51, -2, 3).to(dtype)
    b = ints((B, groups * mb, *hw), 52, -2, 3).to(dtype)
    af = a.float().reshape(B, groups, ma, -1)
    bf = b.float().reshape(B, groups, mb, -1)
    ref = torch.einsum("bgin,bgjn->bgij", af, bf).reshape(B * groups, ma, mb)
    out, ss = o.gram(a.to(DEV), b.to(DEV), groups, False, True)
    assert torch.equal(out.cpu(), ref)
    ssr = torch.cat([af.pow(2).sum(-1), bf.pow(2).sum(-1)], -1).reshape(B * groups, ma + mb)
    assert torch.equal(ss.cpu(), ssr)
    if groups == 1:
        outb = o.gram(a.to(DEV), b.to(DEV), 1, True)
        assert torch.equal(outb.cpu(), ref.sum(0, keepdim=True))


@pytest.mark.parametrize("dtype", DT)
def test_gram_long_reduction_random(dtype):
    """split over the pixel axis across many workgroups (64x64 image = 4096 pixels)."""
    o = ops()
    a = rnd((2, 144, 64, 64), 61).to(dtype)
    b = rnd((2, 48, 64, 64), 62).to(dtype)
    ref = torch.einsum("bin,bjn->ij", a.double().flatten(2), b.double().flatten(2))[None]
    out = o.gram(a.to(DEV), b.to(DEV), 1, True)
    assert rel(out, ref) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("ma,mb,B,hw", [(48, 48, 4, (16, 64)), (144, 48, 5, (16, 64)), (576, 192, 3, (16, 64)), (300, 200, 2, (8, 64)),
                                        (96, 255, 32, (4, 64)), (130, 70, 3, (24, 64)), (1020, 192, 2, (16, 64)), (90, 250, 2, (8, 72)),
                                        (700, 384, 2, (8, 64)), (192, 510, 2, (8, 64)), (384, 1021, 2, (4, 64)), (150, 300, 3, (8, 72))])
def test_gram_batch_fold_exact_on_integers(monkeypatch, dtype, ma, mb, B, hw):
    """Weight-gradient Grams (sum over the batch) with the images chained along the contraction axis: workgroup pixel ranges that
    cross image boundaries, both kernels (LDS-staged and streaming), against the per-image form and the host.  MI_GRAM_FOLD: 0 =
    never, unset = where the planner finds it pays (short planes / half-empty chip), 2 = wherever an image is whole chunks."""
    o = ops()
    a = ints((B, ma, *hw), 151, -2, 3).to(dtype)
    b = ints((B, mb, *hw), 152, -2, 3).to(dtype)
    ref = torch.einsum("bin,bjn->ij", a.float().flatten(2), b.float().flatten(2))[None]
    for mode in ("0", None, "2"):
        if mode is None:
            monkeypatch.delenv("MI_GRAM_FOLD", raising=False)
        else:
            monkeypatch.setenv("MI_GRAM_FOLD", mode)
        out = o.gram(a.to(DEV), b.to(DEV), 1, True)
        assert torch.equal(out.cpu(), ref), (mode, float((out.cpu() - ref).abs().max()))
        monkeypatch.setenv("MI_GRAM_RECT", "0")                 # square 128 x 128 tiles instead of 96 x 256 / 128 x 192
        out = o.gram(a.to(DEV), b.to(DEV), 1, True)
        monkeypatch.delenv("MI_GRAM_RECT")
        assert torch.equal(out.cpu(), ref), ("square", mode)
        monkeypatch.setenv("MI_GRAM_STREAM_ALL", "1")
        out = o.gram(a.to(DEV), b.to(DEV), 1, True)
        monkeypatch.delenv("MI_GRAM_STREAM_ALL")
        assert torch.equal(out.cpu(), ref), ("stream", mode)


# --------------------------------------------------------------------------- AdamW / L1
def test_adamw_matches_torch():
    o = ops()
    n = 10007
    p = rnd((n,), 71)
    g = rnd((n,), 72)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=2e-4)
    pd, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pt.grad = g * step
        opt.step()
        o.adamw_step(pd, (g * step).to(DEV), m, v, 2e-4, step)
    assert rel(pd, pt.detach()) < 1e-6


@pytest.mark.parametrize("dtype", DT)
def test_l1_loss(dtype):
    o = ops()
    a, b = rnd((3, 3, 17, 19), 81).to(dtype), rnd((3, 3, 17, 19), 82).to(dtype)
    loss, da = o.l1_loss(a.to(DEV), b.to(DEV))
    ref = (a.double() - b.double()).abs().mean()
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    assert rel(da, torch.sign(a.double() - b.double()) / a.numel()) < 1e-2


# --------------------------------------------------------------------------- pointwise GEMM, full descriptor
def _pw_raw(x1, x2, w, bias, res, M, groups, w_per_image, transposed):
    """Full mi_pw_desc call: x [B, groups*K, H, W] split head-major into groups; w [(B,) groups, M, K] or its transpose."""
    import ctypes as C
    from image_restoration_amd import _lib as L, ops as o
    B, CK, H, W = x1.shape
    N = H * W
    K1 = CK // groups
    K2 = 0 if x2 is None else x2.shape[1] // groups
    K = K1 + K2
    y = torch.empty((B, groups * M, H, W), dtype=x1.dtype, device=x1.device)
    d = L.PwDesc()
    d.x1, d.x1_bs, d.x1_gs, d.k1 = x1.data_ptr(), CK * N, K1 * N, K1
    if x2 is not None:
        d.x2, d.x2_bs, d.x2_gs, d.k2 = x2.data_ptr(), x2.shape[1] * N, K2 * N, K2
    d.w = w.data_ptr()
    d.w_gs = M * K
    d.w_bs = groups * M * K if w_per_image else 0
    d.w_sm, d.w_sk = (1, M) if transposed else (K, 1)
    if bias is not None:
        d.bias, d.bias_gs = bias.data_ptr(), M
    if res is not None:
        d.r, d.r_bs, d.r_gs = res.data_ptr(), groups * M * N, M * N
    d.y, d.y_bs, d.y_gs = y.data_ptr(), groups * M * N, M * N
    d.m, d.n, d.batch, d.groups, d.dtype = M, N, B, groups, o._dt(x1)
    o.pw_gemm_desc(d, x1.device)
    return y


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("M,K1,K2,groups,per_image,hw", [
    (144, 48, 0, 1, False, (16, 64)), (254, 48, 0, 1, False, (16, 64)), (48, 127, 0, 1, False, (16, 64)),
    (48, 144, 0, 1, False, (8, 64)), (127, 48, 0, 1, False, (8, 64)), (48, 254, 0, 1, False, (8, 64)),
    (288, 96, 0, 1, False, (8, 64)), (96, 255, 0, 1, False, (8, 64)), (96, 288, 0, 1, False, (8, 64)),
    (96, 510, 0, 1, False, (8, 64)), (510, 96, 0, 1, False, (8, 64)), (48, 48, 48, 2, True, (16, 64)),
    (96, 96, 96, 1, True, (8, 72)), (37, 20, 13, 3, True, (5, 7)), (48, 48, 0, 1, True, (9, 11))])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dma", [False, True])
def test_pw_full_descriptor_exact_on_integers(monkeypatch, dma, dtype, transposed, M, K1, K2, groups, per_image, hw):
    """The full mi_pw_desc surface: K 48..510, both weight orientations, two K-panels, head groups, per-image weights,
    ragged pixel counts; register-staged kernel and (dma=True, aligned shapes) the opt-in LDS-DMA ring kernel.
    Integer data: exact up to the final rounding of the store."""
    if dma:
        monkeypatch.setenv("MI_PW_DMA", "1")
    else:
        monkeypatch.delenv("MI_PW_DMA", raising=False)
    B = 2
    K = K1 + K2
    x1 = ints((B, groups * K1, *hw), 91).to(dtype)
    x2 = ints((B, groups * K2, *hw), 92).to(dtype) if K2 else None
    wshape = (B if per_image else 1, groups, M, K)
    w = ints(wshape, 93, -2, 3)
    bias = ints((groups, M), 94)
    res = ints((B, groups * M, *hw), 95).to(dtype)
    xs = x1.float().reshape(B, groups, K1, -1)
    if K2:
        xs = torch.cat([xs, x2.float().reshape(B, groups, K2, -1)], 2)
    ref = torch.einsum("bgmk,bgkn->bgmn", w.expand(B, -1, -1, -1), xs) + bias[None, :, :, None]
    ref = (ref.reshape(B, groups * M, *hw) + res.float()).to(dtype).float()
    warg = w.transpose(-1, -2).contiguous() if transposed else w
    y = _pw_raw(x1.to(DEV), None if x2 is None else x2.to(DEV), warg.to(DEV), bias.to(DEV), res.to(DEV), M, groups,
                per_image, transposed)
    assert torch.equal(y.float().cpu(), ref), f"max diff {(y.float().cpu() - ref).abs().max()}"


@pytest.mark.parametrize("epilogue", ["none", "bias"])
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("M,K,groups,per_image", [(510, 96, 1, False), (254, 48, 1, False), (144, 40, 1, False), (96, 510, 1, False),
                                                  (48, 127, 1, False), (70, 200, 1, False), (96, 96, 1, True), (48, 48, 2, True),
                                                  (130, 33, 1, False)])
def test_pw_wave_forms_without_residual(monkeypatch, epilogue, transposed, M, K, groups, per_image):
    """The wave-owned GEMM forms (64-pixel-aligned bf16 rows) store through a bf16 patch when there is no residual: wide outputs
    (all channels walked 32 at a time), narrow outputs over long K, odd K tails, M tails inside a 32-channel pair, bias alone.
    Integer data -> exact; the chunked kernel (MI_PW_WAVE=0) must give the same bits."""
    B, hw = 2, (16, 64)
    dtype = torch.bfloat16
    x1 = ints((B, groups * K, *hw), 191).to(dtype)
    w = ints((B if per_image else 1, groups, M, K), 193, -2, 3)
    bias = ints((groups, M), 194) if epilogue == "bias" else None
    ref = torch.einsum("bgmk,bgkn->bgmn", w.expand(B, -1, -1, -1), x1.float().reshape(B, groups, K, -1))
    if bias is not None:
        ref = ref + bias[None, :, :, None]
    ref = ref.reshape(B, groups * M, *hw).to(dtype).float()
    warg = w.transpose(-1, -2).contiguous() if transposed else w
    outs = []
    for wave in ("1", "0"):
        monkeypatch.setenv("MI_PW_WAVE", wave)
        y = _pw_raw(x1.to(DEV), None, warg.to(DEV), None if bias is None else bias.to(DEV), None, M, groups, per_image, transposed)
        outs.append(y.float().cpu())
    assert torch.equal(outs[0], ref), f"max diff {(outs[0] - ref).abs().max()}"
    assert torch.equal(outs[1], ref)
    if per_image:                                       # A/B switch: per-image weights staged from fp32 inside the wave-owned GEMM
        monkeypatch.setenv("MI_PW_WAVE", "1")
        monkeypatch.setenv("MI_PW_DIRECT", "1")
        y = _pw_raw(x1.to(DEV), None, warg.to(DEV), None if bias is None else bias.to(DEV), None, M, groups, per_image, transposed)
        assert torch.equal(y.float().cpu(), ref)


def test_pw_wave_forms_fuzz_against_chunked(monkeypatch):
    """Seeded sweep over the descriptor space the wave-owned forms accept (M 1..600, K 1..520 in one or two panels, head groups,
    per-image weights, bias / residual on or off, odd tile counts): integer data, so the wave-owned kernels, the chunked kernel and the
    host result must agree exactly."""
    import random
    rng = random.Random(20260101)
    dtype = torch.bfloat16
    for case in range(40):
        M = rng.choice([1, 7, 16, 24, 48, 49, 64, 65, 96, 97, 127, 130, 192, 255, 300, 510, 600])
        K = rng.choice([1, 5, 32, 33, 48, 64, 96, 97, 127, 160, 192, 255, 288, 384, 510, 520])
        two = rng.random() < 0.3 and K >= 2
        K1 = rng.randint(1, K - 1) if two else K
        K2 = K - K1
        groups = rng.choice([1, 1, 1, 2, 3]) if M * K <= 130 * 130 else 1
        per_image = rng.random() < 0.3
        transposed = rng.random() < 0.5
        hw = rng.choice([(8, 8), (1, 64), (3, 64), (8, 72), (16, 64)])        # 1, 1, 3, 9, 16 tiles of 64 pixels
        use_bias, use_res = rng.random() < 0.5, rng.random() < 0.5
        B = 2
        x1 = ints((B, groups * K1, *hw), 1000 + case).to(dtype)
        x2 = ints((B, groups * K2, *hw), 2000 + case).to(dtype) if K2 else None
        w = ints((B if per_image else 1, groups, M, K), 3000 + case, -2, 3)
        bias = ints((groups, M), 4000 + case) if use_bias else None
        res = ints((B, groups * M, *hw), 5000 + case).to(dtype) if use_res else None
        xs = x1.float().reshape(B, groups, K1, -1)
        if K2:
            xs = torch.cat([xs, x2.float().reshape(B, groups, K2, -1)], 2)
        ref = torch.einsum("bgmk,bgkn->bgmn", w.expand(B, -1, -1, -1), xs)
        if bias is not None:
            ref = ref + bias[None, :, :, None]
        ref = ref.reshape(B, groups * M, *hw)
        if res is not None:
            ref = ref + res.float()
        ref = ref.to(dtype).float()
        warg = w.transpose(-1, -2).contiguous() if transposed else w
        for wave in ("1", "0"):
            monkeypatch.setenv("MI_PW_WAVE", wave)
            y = _pw_raw(x1.to(DEV), None if x2 is None else x2.to(DEV), warg.to(DEV), None if bias is None else bias.to(DEV),
                        None if res is None else res.to(DEV), M, groups, per_image, transposed)
            assert torch.equal(y.float().cpu(), ref), (case, wave, M, K1, K2, groups, per_image, transposed, hw, use_bias, use_res)


@pytest.mark.parametrize("epilogue", ["none", "bias", "res", "bias+res"])
@pytest.mark.parametrize("M,K1,K2,per_image,transposed,hw", [
    (576, 192, 0, False, False, (16, 64)), (1020, 192, 0, False, False, (8, 72)), (300, 130, 0, False, True, (3, 64)),
    (256, 128, 0, True, False, (8, 72)), (600, 97, 0, False, False, (1, 64)), (510, 100, 92, False, True, (16, 64)),
    (2042, 160, 0, False, False, (4, 64)),
    # K > 192 stays on the streaming form: same expectations
    (1152, 384, 0, False, False, (16, 64)), (300, 200, 0, True, True, (3, 64))])
def test_pw_xwide_form_exact_on_integers(monkeypatch, epilogue, M, K1, K2, per_image, transposed, hw):
    """The X-resident / W-streamed wave form (bf16, K = 97 .. 192, M >= 256: the C = 192 level's qkv and project_in): every
    output-channel tile walked behind one barrier, ragged M inside the last 64-row tile, K tails inside a 32-k chunk, two K
    panels, per-image weights, waves past the plane (9 tiles on 8-wave workgroups), the M split that fills the chip on small
    batches.  Integer data -> exact against the host; MI_PW_XWIDE=0 (the streaming form it replaces) must give the same bits."""
    B, dtype = 2, torch.bfloat16
    K = K1 + K2
    x1 = ints((B, K1, *hw), 291).to(dtype)
    x2 = ints((B, K2, *hw), 292).to(dtype) if K2 else None
    w = ints((B if per_image else 1, 1, M, K), 293, -2, 3)
    bias = ints((1, M), 294) if "bias" in epilogue else None
    res = ints((B, M, *hw), 295).to(dtype) if "res" in epilogue else None
    xs = x1.float().reshape(B, 1, K1, -1)
    if K2:
        xs = torch.cat([xs, x2.float().reshape(B, 1, K2, -1)], 2)
    ref = torch.einsum("bgmk,bgkn->bgmn", w.expand(B, -1, -1, -1), xs)
    if bias is not None:
        ref = ref + bias[None, :, :, None]
    ref = ref.reshape(B, M, *hw)
    if res is not None:
        ref = ref + res.float()
    ref = ref.to(dtype).float()
    warg = w.transpose(-1, -2).contiguous() if transposed else w
    for xwide in ("1", "0"):
        monkeypatch.setenv("MI_PW_XWIDE", xwide)
        y = _pw_raw(x1.to(DEV), None if x2 is None else x2.to(DEV), warg.to(DEV), None if bias is None else bias.to(DEV),
                    None if res is None else res.to(DEV), M, 1, per_image, transposed)
        assert torch.equal(y.float().cpu(), ref), (xwide, float((y.float().cpu() - ref).abs().max()))
        if per_image:                                   # A/B switch: per-image weights staged from fp32 inside the GEMM
            monkeypatch.setenv("MI_PW_DIRECT", "1")
            y = _pw_raw(x1.to(DEV), None if x2 is None else x2.to(DEV), warg.to(DEV), None if bias is None else bias.to(DEV),
                        None if res is None else res.to(DEV), M, 1, per_image, transposed)
            monkeypatch.delenv("MI_PW_DIRECT")
            assert torch.equal(y.float().cpu(), ref), ("direct", xwide)
