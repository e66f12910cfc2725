"""GPU parity of the implicit-GEMM dense 3x3 convolution (csrc/conv3x3.hip: the U-Net glue convs of Restormer.py:156-189,243,281)
against torch.nn.functional.conv2d / conv_transpose2d in float64 on the CPU (the same bf16-rounded inputs and weights, so the
bound is the bf16 rounding of the output plus fp32 accumulation order: 1e-2 of the largest output magnitude)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


CASES = [
    # B, M (out), K (in), H, W, bias, residual
    (2, 48, 3, 32, 64, False, False),        # OverlapPatchEmbed: K = 3 (one chunk, 13 zero channels)
    (2, 3, 96, 16, 64, True, True),          # output conv + input residual: M = 3 (one m-tile), bias
    (1, 24, 48, 16, 128, False, False),      # Downsample level 1: two 64-pixel column tiles
    (2, 48, 96, 24, 32, False, False),       # 32-pixel rows (TWP = 32), rows not a multiple of the tile
    (2, 96, 192, 16, 16, True, False),       # 16-pixel rows (TWP = 16)
    (1, 768, 384, 8, 32, False, False),      # Upsample at the latent level: 12 output-channel tiles, 24 chunks
    (1, 384, 192, 16, 64, False, False),
    (3, 40, 24, 10, 24, True, True),         # ragged everything: K = 24 (half-empty chunk), M = 40, W = 24, H = 10
    (1, 20, 30, 7, 72, False, True),         # W = 72: a full and a partial column tile
    (2, 192, 96, 12, 8, False, False),       # W = 8: a single vector per row
]


@pytest.mark.parametrize("B,M,K,H,W,bias,res", CASES)
def test_conv3x3_forward_vs_fp64(B, M, K, H, W, bias, res):
    from image_restoration_amd import ops
    g = torch.Generator().manual_seed(100 + M + K + W)
    x = torch.randn((B, K, H, W), generator=g).to(torch.bfloat16)
    w = (torch.randn((M, K, 3, 3), generator=g) / (3.0 * K ** 0.5)).to(torch.bfloat16).float()   # bf16-representable fp32 weights
    bv = torch.randn(M, generator=g) if bias else None
    rv = torch.randn((B, M, H, W), generator=g).to(torch.bfloat16) if res else None
    ref = F.conv2d(x.double(), w.double(), bv.double() if bias else None, padding=1)
    if res:
        ref = ref + rv.double()
    xd = x.to(DEV)
    assert ops.conv3x3_ok(xd)
    got = ops.conv3x3(xd, w.to(DEV), bv.to(DEV) if bias else None, rv.to(DEV) if res else None)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    e = rel(got, ref)
    assert e < 1e-2, e


@pytest.mark.parametrize("B,M,K,H,W", [(2, 48, 24, 16, 128), (1, 96, 48, 24, 32), (2, 384, 768, 8, 16), (2, 96, 3, 16, 64),
                                       (1, 30, 20, 9, 40)])
def test_conv3x3_data_gradient_vs_fp64(B, M, K, H, W):
    """transpose=True: the op is the data gradient of a conv whose weight is [K (its outputs), M (its inputs), 3, 3]."""
    from image_restoration_amd import ops
    g = torch.Generator().manual_seed(200 + M + K + W)
    dy = torch.randn((B, K, H, W), generator=g).to(torch.bfloat16)
    w = (torch.randn((K, M, 3, 3), generator=g) / (3.0 * K ** 0.5)).to(torch.bfloat16).float()
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    got = ops.conv3x3(dy.to(DEV), w.to(DEV), transpose=True)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    e = rel(got, ref)
    assert e < 1e-2, e


def test_conv3x3_channel_slices_and_refusals():
    from image_restoration_amd import ops
    g = torch.Generator().manual_seed(7)
    big = torch.randn((2, 80, 16, 64), generator=g).to(torch.bfloat16).to(DEV)
    x = big[:, 16:64]                                        # a channel slice: batch stride 80 planes
    w = (torch.randn((24, 48, 3, 3), generator=g) / 20).to(DEV)
    out_big = torch.zeros((2, 40, 16, 64), dtype=torch.bfloat16, device=DEV)
    ops.conv3x3(x, w, out=out_big[:, 8:32])
    ref = F.conv2d(x.float().cpu().double(), w.cpu().to(torch.bfloat16).double(), padding=1)
    assert rel(out_big[:, 8:32], ref) < 1e-2
    assert float(out_big[:, :8].abs().max()) == 0.0 and float(out_big[:, 32:].abs().max()) == 0.0
    assert not ops.conv3x3_ok(torch.zeros((1, 8, 8, 12), dtype=torch.bfloat16, device=DEV))     # W % 8 != 0
    assert not ops.conv3x3_ok(torch.zeros((1, 8, 8, 16), dtype=torch.float32, device=DEV))      # fp32 stays on the exact forms
    with pytest.raises(ValueError):
        ops.conv3x3(big[:, :48], torch.zeros((24, 40, 3, 3), device=DEV))


@pytest.mark.parametrize("B,M,K,H,W", [(2, 48, 3, 32, 64), (2, 3, 96, 16, 64), (2, 24, 48, 16, 128), (2, 96, 48, 24, 32),
                                       (2, 192, 96, 16, 16), (1, 768, 384, 8, 32), (3, 40, 24, 10, 24), (1, 20, 30, 7, 72)])
def test_conv3x3_weight_gradient_vs_fp64(B, M, K, H, W):
    """mi_conv3x3_wgrad against autograd of F.conv2d in float64 on the same bf16 operands; then accumulate on top of it."""
    from image_restoration_amd import ops
    g = torch.Generator().manual_seed(300 + M + K + W)
    x = torch.randn((B, K, H, W), generator=g).to(torch.bfloat16)
    dy = torch.randn((B, M, H, W), generator=g).to(torch.bfloat16)
    w = torch.zeros((M, K, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=1).backward(dy.double())
    got = ops.conv3x3_wgrad(dy.to(DEV), x.to(DEV))
    torch.cuda.synchronize()
    e = rel(got, w.grad)
    assert e < 2e-5 * (B * H * W) ** 0.5 + 1e-6, e        # exact bf16 products, fp32 accumulation over B*H*W terms
    base = got.clone()
    ops.conv3x3_wgrad(dy.to(DEV), x.to(DEV), got, accumulate=True)
    assert rel(got, 2 * base) < 1e-6


@pytest.mark.parametrize("cin,cout,hw,res", [(3, 48, (32, 64), False), (96, 3, (16, 64), True), (48, 24, (32, 32), False),
                                             (96, 192, (16, 16), False)])
def test_glue_conv_module_both_routes(cin, cout, hw, res, monkeypatch):
    """restormer._conv2d (the door the models use) on the implicit-GEMM route (default) and on the im2col route
    (MI_NO_CONV3_IMPLICIT=1): output, dx, dW, db and the residual gradient against F.conv2d in float64."""
    import image_restoration_amd as m
    from image_restoration_amd import restormer
    g = torch.Generator().manual_seed(400 + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=True)
    x0 = torch.randn((2, cin) + hw, generator=g).to(torch.bfloat16)
    r0 = torch.randn((2, cout) + hw, generator=g).to(torch.bfloat16) if res else None
    cot = torch.randn((2, cout) + hw, generator=g).to(torch.bfloat16)
    xr = x0.double().requires_grad_(True)
    rr = r0.double().requires_grad_(True) if res else None
    wr, br = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1) + (rr if res else 0)
    ref.backward(cot.double())
    for route in ("implicit", "im2col"):
        if route == "im2col":
            monkeypatch.setenv("MI_NO_CONV3_IMPLICIT", "1")
        else:
            monkeypatch.delenv("MI_NO_CONV3_IMPLICIT", raising=False)
        m.reload_env()
        c = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=True).to(DEV)
        c.load_state_dict(conv.state_dict())
        x = x0.to(DEV).requires_grad_(True)
        r = r0.to(DEV).requires_grad_(True) if res else None
        y = restormer._conv2d(x, c, r)
        y.backward(cot.to(DEV))
        assert rel(y, ref) < 1.5e-2, (route, rel(y, ref))
        assert rel(x.grad, xr.grad) < 1.5e-2, (route, rel(x.grad, xr.grad))
        assert rel(c.weight.grad, wr.grad) < 1.5e-2, (route, rel(c.weight.grad, wr.grad))
        assert rel(c.bias.grad, br.grad) < 1.5e-2, route
        if res:
            assert rel(r.grad, rr.grad) < 1e-6, route


# ------------------------------------------------------------------------------------------------ LDS-tiled 1x1 GEMM for deep K (csrc/pw_lds.hip)
@pytest.mark.parametrize("B,M,K,hw,tr,bias,res", [
    (2, 1152, 384, (32, 32), False, False, False),      # latent qkv
    (2, 2042, 384, (32, 32), False, True, False),       # latent project_in (M = 2042: ragged last tile), bias
    (2, 384, 1021, (32, 32), False, False, True),       # latent project_out: K = 1021 (ragged last chunk), residual
    (1, 1020, 192, (64, 64), False, False, False),      # level 3 project_in
    (2, 192, 510, (16, 32), False, True, True),         # level 3 project_out on a 512-pixel plane
    (2, 384, 2042, (32, 32), True, False, False),       # backward of project_in: W^T, K = 2042
    (1, 1021, 384, (20, 24), True, False, False),       # backward of project_out; 480-pixel plane (partial pixel tile)
    (3, 128, 129, (16, 16), False, False, False),       # the smallest shape the kernel takes
])
def test_deep_1x1_gemm_vs_fp64(B, M, K, hw, tr, bias, res, monkeypatch):
    """ops.conv1x1 on the shapes mi_pw_gemm hands to the LDS-tiled kernel (bf16, K > 128, M >= 128), against fp64 on the same
    bf16-rounded operands, and against the kernels it replaces (MI_NO_PW_LDS=1): both within the bf16 output rounding."""
    import image_restoration_amd as m
    from image_restoration_amd import ops
    g = torch.Generator().manual_seed(500 + M + K)
    x = torch.randn((B, K) + hw, generator=g).to(torch.bfloat16)
    w = (torch.randn((K, M) if tr else (M, K), generator=g) / K ** 0.5).to(torch.bfloat16).float()
    bv = torch.randn(M, generator=g) if bias else None
    rv = torch.randn((B, M) + hw, generator=g).to(torch.bfloat16) if res else None
    wm = w.t() if tr else w
    ref = torch.einsum("mk,bkhw->bmhw", wm.double(), x.double())
    if bias:
        ref = ref + bv.double().view(1, -1, 1, 1)
    if res:
        ref = ref + rv.double()
    outs = {}
    monkeypatch.setenv("MI_PW_LDS", "all")                 # every covered shape, not only the ones where it is the faster kernel
    for mode in ("lds", "old"):
        if mode == "old":
            monkeypatch.setenv("MI_NO_PW_LDS", "1")
        else:
            monkeypatch.delenv("MI_NO_PW_LDS", raising=False)
        m.reload_env()
        outs[mode] = ops.conv1x1(x.to(DEV), w.to(DEV), bv.to(DEV) if bias else None, rv.to(DEV) if res else None, transposed=tr)
        torch.cuda.synchronize()
        assert rel(outs[mode], ref) < 1e-2, (mode, rel(outs[mode], ref))
    assert rel(outs["lds"], outs["old"].float()) < 1e-2
