"""GPU parity of the fp8 (OCP e4m3) MFMA projections - BASELINE configs[4] "CDNA4 fp8 MFMA projections" (mi_pw_desc.f8,
mi_mdta_fwd_f8, mi_gdfn_fwd_f8, restormer.fp8_calibrate / fp8_projections).

The reference has no fp8 path ("No bf16, no fp8 anywhere"), so the oracle for the arithmetic is the definition in
include/mi_restore.h restated on the CPU with torch's float8_e4m3fn: y = bf16( (e4m3(x / sx) . e4m3(w / sw)) * sx * sw + bias +
residual ), fp32 accumulation, w taken from the packed bf16 weight image.  The GEMM must match that to the output's bf16 rounding.  Against the bf16 path the fp8 path
carries the e4m3 rounding of its operands (3 mantissa bits: ~3.5 % per element, measured 3.3 % on a K = 32 product by
tools/microbench/f8_probe.hip); the whole-block / whole-network bars below are PSNR bars and are stated where they are used."""
import pytest
import torch

from oracle import restormer_ref as R
from oracle.fixtures import seeded_input

pytestmark = pytest.mark.gpu
DEV = "cuda"
F8 = torch.float8_e4m3fn


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def q8(t, scale):
    """e4m3(t / scale), saturating at +-448, as fp32 (the operand the MFMA sees)."""
    return (t.float() / scale).clamp(-448.0, 448.0).to(F8).float()


def psnr(a, b):
    return float(10 * torch.log10(1.0 / torch.mean((a.double() - b.double()) ** 2)))


def f8_ref(x, w, sx, sw, bias=None, residual=None):
    B, K, H, W = x.shape
    # (the kernel converts the PACKED weight image, i.e. the fp32 weight rounded to bf16 first)
    wq = q8(w.cpu().to(torch.bfloat16), sw)
    y = torch.einsum("mk,bkn->bmn", wq.double(), q8(x.cpu(), sx).reshape(B, K, -1).double()) * (sx * sw)
    if bias is not None:
        y = y + bias.cpu().double()[None, :, None]
    y = y.reshape(B, -1, H, W)
    if residual is not None:
        y = y + residual.cpu().double()
    return y


GEMM_CASES = [
    # M, K, shape (B, H, W), bias, residual        kernel form
    (144, 48, (2, 16, 64), False, False),        # X-resident (qkv at C = 48)
    (288, 96, (1, 32, 64), True, False),         # X-resident, three k-chunks, bias
    (254, 48, (2, 16, 64), False, False),        # X-resident, ragged M (project_in at C = 48)
    (48, 48, (2, 32, 64), False, True),          # streaming, one 48-row tile, residual
    (48, 127, (2, 16, 64), False, True),         # streaming, ragged K (project_out at C = 48)
    (96, 96, (1, 32, 64), True, True),           # streaming, 96-row tile
    (192, 192, (2, 16, 64), False, True),        # streaming, two m-tiles (level 3)
    (384, 384, (1, 16, 64), False, False),       # streaming, K = 384 (latent)
    (576, 192, (1, 16, 64), False, False),       # qkv at C = 192
]


@pytest.mark.parametrize("M,K,shape,bias,res", GEMM_CASES)
def test_pw_gemm_f8_matches_the_e4m3_definition(M, K, shape, bias, res):
    from image_restoration_amd import ops
    B, H, W = shape
    x = (seeded_input((B, K, H, W), 900 + M + K) * 1.3).to(torch.bfloat16).to(DEV)
    w = (seeded_input((M, K), 901 + M) * 0.2).float().to(DEV)
    b = seeded_input((M,), 902).float().to(DEV) if bias else None
    r = seeded_input((B, M, H, W), 903).to(torch.bfloat16).to(DEV) if res else None
    sx, sw = 2.0 ** -5, 2.0 ** -8                     # |x| <= ~6, |w| <= ~1: everything inside +-448 after scaling
    assert float(x.abs().max()) / sx <= 448 and float(w.abs().max()) / sw <= 448
    got = ops.conv1x1(x, w, b, r, f8=(sx, sw))
    ref = f8_ref(x, w, sx, sw, b, r)
    # the product of two e4m3 values is exact in fp32 and K <= 384 partial sums keep ~7 digits: what is left is the bf16
    # rounding of the stored output (2^-9 relative to the element; the bound is relative to the largest output)
    assert rel(got, ref) < 4e-3, rel(got, ref)
    # and the fp8 result differs from the bf16 GEMM by the operand rounding only (a few per cent of the output scale)
    full = ops.conv1x1(x, w, b, r)
    assert 1e-4 < rel(got, full.float()) < 8e-2, rel(got, full.float())


def test_pw_gemm_f8_saturates_instead_of_nan():
    """A scale that is too small must clamp at +-448 (MODE.FP16_OVFL), not turn the tile into NaN."""
    from image_restoration_amd import ops
    x = (seeded_input((1, 48, 16, 64), 77) * 50).to(torch.bfloat16).to(DEV)       # |x| / sx far beyond 448
    w = (seeded_input((48, 48), 78) * 0.2).float().to(DEV)
    sx, sw = 2.0 ** -6, 2.0 ** -9
    got = ops.conv1x1(x, w, None, None, f8=(sx, sw))
    assert torch.isfinite(got.float()).all()
    assert rel(got, f8_ref(x, w, sx, sw)) < 4e-3


def test_pw_gemm_f8_rejects_uncovered_forms():
    from image_restoration_amd import ops
    x32 = seeded_input((1, 48, 16, 64), 5).float().to(DEV)
    w = seeded_input((48, 48), 6).float().to(DEV)
    with pytest.raises(RuntimeError, match="fp8"):
        ops.conv1x1(x32, w, f8=(1.0, 1.0))                                        # fp32 activations
    xr = seeded_input((1, 48, 10, 10), 7).to(torch.bfloat16).to(DEV)             # ragged plane: not a wave-owned form
    with pytest.raises(RuntimeError, match="fp8"):
        ops.conv1x1(xr, w, f8=(1.0, 1.0))
    xb = seeded_input((1, 48, 16, 64), 8).to(torch.bfloat16).to(DEV)
    with pytest.raises(RuntimeError, match="scales"):
        ops.conv1x1(xb, w, f8=(0.0, 1.0))


HALF_CASES = [
    # C, heads, f, bias, LN kind, shape
    (48, 1, 2.66, False, "WithBias", (2, 48, 32, 64)),
    (96, 2, 2.66, True, "WithBias", (1, 96, 32, 64)),
    (96, 1, 2.66, False, "BiasFree", (1, 96, 16, 64)),
    (128, 4, 2.66, False, "WithBias", (1, 128, 16, 64)),     # LayerNorm inside the W-streamed X-resident form
    (192, 4, 2.66, False, "WithBias", (1, 192, 16, 64)),     # no LayerNorm head at this width: norm runs as its own kernel
]


@pytest.mark.parametrize("c,heads,f,bias,kind,shape", HALF_CASES)
def test_half_blocks_on_fp8_operands(c, heads, f, bias, kind, shape):
    """mi_mdta_fwd_f8 / mi_gdfn_fwd_f8 against (i) the same chain assembled from single fp8 GEMM calls and the bf16 kernels
    around them - identical arithmetic, so a few bf16 ulps - and (ii) the bf16 half-block: the fp8 branch output stays within
    8 % of the branch's largest magnitude (operand rounding of two chained projections)."""
    from image_restoration_amd import ops
    sd = R.make_block_state(c, heads, f, bias, kind, seed=61 + c)
    dev = lambda k: sd[k].to(DEV).float().contiguous() if k in sd else None
    att = (dev("attn.temperature"), dev("attn.qkv.weight"), dev("attn.qkv.bias"), dev("attn.qkv_dwconv.weight"),
           dev("attn.qkv_dwconv.bias"), dev("attn.project_out.weight"), dev("attn.project_out.bias"))
    ffn = tuple(dev(k) for k in ("ffn.project_in.weight", "ffn.project_in.bias", "ffn.dwconv.weight", "ffn.dwconv.bias",
                                 "ffn.project_out.weight", "ffn.project_out.bias"))
    x = (seeded_input(shape, 8200 + c) * 1.7 + 0.4).to(DEV).to(torch.bfloat16)
    wb = kind == "WithBias"
    n1 = (dev("norm1.body.weight"), dev("norm1.body.bias"))
    hidden = ffn[4].shape[1]
    with_ln = ops.mdta_fwd_ln_ok(x, heads, 3)
    assert with_ln == (c <= 128)
    assert ops.mdta_fwd_f8_ok(x, heads, 3, with_ln) and ops.gdfn_fwd_f8_ok(x, hidden, 3, with_ln)
    xn, _, _ = ops.ln_fwd(x, n1[0], n1[1], wb, want_stats=False)
    # scales as restormer.fp8_calibrate derives them
    from image_restoration_amd.restormer import _f8_pow2
    v = ops.dwconv_fwd(ops.conv1x1(xn, att[1], att[2]), att[3], att[4])[:, 2 * c:]
    g = ops.dwconv_gate_fwd(ops.conv1x1(xn, ffn[0], ffn[1]), ffn[2], ffn[3], want_y=False)[1]
    wo_bound = float(att[5].abs().reshape(c, heads, c // heads).sum(-1).max())
    s_att = (_f8_pow2(4 * float(xn.abs().max())), _f8_pow2(float(att[1].abs().max())), _f8_pow2(4 * float(v.abs().max())),
             _f8_pow2(wo_bound))
    s_ffn = (_f8_pow2(4 * float(xn.abs().max())), _f8_pow2(float(ffn[0].abs().max())), _f8_pow2(4 * float(g.abs().max())),
             _f8_pow2(float(ffn[4].abs().max())))
    ln = (n1[0], n1[1], False) if with_ln else None
    xin = x if with_ln else xn

    got = ops.mdta_fwd(xin, x, att, heads, False, ln=ln, f8=s_att)
    ref16 = ops.mdta_fwd(xn, x, att, heads, False)[0]
    branch = (ref16.float() - x.float())
    err = float((got.float() - ref16.float()).abs().max() / branch.abs().max())
    assert 0 < err < 8e-2, err

    got = ops.gdfn_fwd(xin, x, ffn, False, ln=ln, f8=s_ffn)
    # (i) the same chain from single calls
    h0 = ops.conv1x1(xn, ffn[0], ffn[1], f8=(s_ffn[0], s_ffn[1]))
    gg = ops.dwconv_gate_fwd(h0, ffn[2], ffn[3], want_y=False)[1]
    chain = ops.conv1x1(gg, ffn[4], ffn[5], x, f8=(s_ffn[2], s_ffn[3]))
    assert rel(got, chain.float()) < (3e-2 if with_ln else 1e-6), rel(got, chain.float())
    # (ii) the bf16 half-block
    ref16 = ops.gdfn_fwd(xn, x, ffn, False)[0]
    branch = (ref16.float() - x.float())
    err = float((got.float() - ref16.float()).abs().max() / branch.abs().max())
    assert 0 < err < 8e-2, err


def test_fp8_network_modes_and_psnr_bar():
    """Restormer (narrow, all four levels) under no_grad: calibrate on the input, then 'attn' and 'all'.  Bars: every projection
    of the selected halves runs on fp8 operands (coverage counters); the outputs stay finite; PSNR of the fp8 output against
    the bf16 output >= 38 dB and the PSNR against the clean target moves by < 0.05 dB (measured at base width on 1024^2:
    profiles/r02_g_fp8_inference_1024.txt)."""
    import image_restoration_amd as m
    from image_restoration_amd import restormer
    torch.manual_seed(3)
    net = m.Restormer(inp_channels=3, out_channels=3, dim=48, num_blocks=[1, 1, 1, 2], num_refinement_blocks=1,
                      heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias").to(DEV)
    clean = torch.rand((2, 3, 128, 128), device=DEV)
    noisy = (torch.clamp(torch.round(clean * 255) + 25 * torch.randn_like(clean), 0, 255) / 255).to(torch.bfloat16)
    with torch.no_grad():
        out16 = net(noisy).float()
        with pytest.raises(RuntimeError, match="fp8_calibrate"):
            restormer.fp8_projections(net, "all")
        restormer.fp8_calibrate(net, [noisy])
        again = net(noisy).float()
        assert torch.equal(again, out16)                      # calibration leaves the bf16 path as it was
        nblocks = len(restormer._blocks(net))
        for mode, want in (("attn", 2 * nblocks), ("all", 4 * nblocks)):
            restormer.fp8_projections(net, mode)
            restormer.F8_COUNTS.update(f8=0, bf16=0)
            out8 = net(noisy).float()
            assert restormer.F8_COUNTS["f8"] == want, (mode, restormer.F8_COUNTS)
            assert restormer.F8_COUNTS["f8"] + restormer.F8_COUNTS["bf16"] == 4 * nblocks
            assert torch.isfinite(out8).all()
            assert not torch.equal(out8, out16)
            p = psnr(out8, out16)
            assert p > 38.0, (mode, p)
            assert abs(psnr(out8, clean) - psnr(out16, clean)) < 0.05
        restormer.fp8_projections(net, None)
        assert torch.equal(net(noisy).float(), out16)


@pytest.mark.parametrize("c,shape", [(48, (2, 48, 32, 64)), (48, (1, 48, 8, 64)), (96, (1, 96, 16, 64))])
def test_fused_gdfn_on_fp8_operands(c, shape):
    """mi_gdfn_fused_fwd_f8 (the one-launch LayerNorm + GDFN half-block, both projections on e4m3 operands) against the bf16 launch
    of the same kernel: the feed-forward branch stays within 12 % of its largest magnitude at the worst element (rms 6-7 % of the rms of the branch), and
    the two fp8 forms agree with each other to the same bound (different operand scales: the fused kernel quantises the
    normalised input and W_in . diag(gamma), the chain LN(y) and W_in)."""
    import math
    from image_restoration_amd import ops
    from image_restoration_amd.restormer import _f8_pow2
    sd = R.make_block_state(c, 1, 2.66, False, "WithBias", seed=71 + c)
    dev = lambda k: sd[k].to(DEV).float().contiguous() if k in sd else None
    ffn = tuple(dev(k) for k in ("ffn.project_in.weight", "ffn.project_in.bias", "ffn.dwconv.weight", "ffn.dwconv.bias",
                                 "ffn.project_out.weight", "ffn.project_out.bias"))
    n2 = (dev("norm2.body.weight"), dev("norm2.body.bias"))
    y = (seeded_input(shape, 8300 + c) * 1.7 + 0.4).to(DEV).to(torch.bfloat16)
    hidden = ffn[4].shape[1]
    assert ops.gdfn_fused_ok(y, hidden, 3)
    pack = ops.gdfn_fused_pack(y, n2[0], n2[1], ffn)
    ref16 = ops.gdfn_fused_fwd(y, pack, hidden, True)[0]
    yn, _, _ = ops.ln_fwd(y, n2[0], n2[1], True, want_stats=False)
    g = ops.dwconv_gate_fwd(ops.conv1x1(yn, ffn[0], ffn[1]), ffn[2], ffn[3], want_y=False)[1]
    wfold = float((ffn[0].reshape(ffn[0].shape[0], -1) * n2[0][None, :]).abs().max())
    x2, w2 = _f8_pow2(4 * float(g.abs().max())), _f8_pow2(float(ffn[4].abs().max()))
    got = ops.gdfn_fused_fwd(y, pack, hidden, True, f8=(_f8_pow2(math.sqrt(c)), _f8_pow2(wfold), x2, w2))[0]
    assert torch.isfinite(got.float()).all()
    branch = (ref16.float() - y.float())
    # (worst element over the tile: 12 % of the largest magnitude of the branch; root-mean-square: 10 % of its rms - measured 6-7 %)
    err = float((got.float() - ref16.float()).abs().max() / branch.abs().max())
    rms = float((got.float() - ref16.float()).pow(2).mean().sqrt() / branch.pow(2).mean().sqrt())
    assert 0 < err < 0.12 and rms < 0.1, (err, rms)
    chain = ops.gdfn_fwd(yn, y, ffn, False, f8=(_f8_pow2(4 * float(yn.abs().max())), _f8_pow2(float(ffn[0].abs().max())), x2, w2))
    err = float((got.float() - chain.float()).abs().max() / branch.abs().max())
    rms = float((got.float() - chain.float()).pow(2).mean().sqrt() / branch.pow(2).mean().sqrt())
    assert err < 0.12 and rms < 0.1, (err, rms)
    with pytest.raises(ValueError):
        ops.gdfn_fused_fwd(y, pack, hidden, True, want_stats=True, f8=(1.0, 1.0, 1.0, 1.0))
