"""GPU parity of AdaIR's frequency modules (csrc/adair.hip, image_restoration_amd/adair.py) through the C-ABI: every kernel
against the fp64 CPU oracle / plain torch autograd on the same seeded inputs, FreModule and the assembled network against the
golden vectors captured from the imported reference (tools/capture_golden_adair.py).
Bounds: fp32 activations 1e-4 forward / 1e-3 gradients of the tensor's largest magnitude; bf16 3e-2 (storage rounding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import adair_ref as A
from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = torch.float64


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("factor", [1, 2, 4, 8])
def test_box_down_is_bilinear_at_level_factors(factor):
    from image_restoration_amd import ops
    img = seeded_input((2, 3, 64, 96), 10 + factor)
    ref = F.interpolate(img.double(), (64 // factor, 96 // factor), mode="bilinear")
    got = ops.box_down(img.to(DEV), 64 // factor, 96 // factor)
    assert rel(got, ref) < 1e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("hw,halves", [((16, 16), None), ((256, 384), [(1, 2), (0, 1), (2, 3)]), ((130, 140), [(1, 1), (1, 1), (1, 1)])])
def test_fre_split_fwd_bwd_vs_fft_oracle(hw, halves, dtype, tol):
    """high = |x - Px|, low = |Px| evaluated as a direct DFT at the rectangle's frequencies, against the reference's fft2 / mask /
    ifft2 (oracle.adair_ref.fre_split) - forward and the gradient through both magnitudes; per-sample rectangles, one empty."""
    from image_restoration_amd import adair
    B, C = 3, 5
    feat = seeded_input((B, C, *hw), 30 + hw[0]).to(dtype)
    half = None if halves is None else torch.tensor(halves, dtype=torch.int32)
    xr = feat.double().requires_grad_(True)
    hr, lr = A.fre_split(xr, torch.zeros((B, 2), dtype=torch.long) if half is None else half.long())
    ch, cl = seeded_input(tuple(hr.shape), 31).double(), seeded_input(tuple(hr.shape), 32).double()
    (hr * ch + lr * cl).sum().backward()
    x = feat.to(DEV).requires_grad_(True)
    high, low = adair._FreSplitFn.apply(x, None if half is None else half.to(DEV))
    (high.float() * ch.float().to(DEV) + low.float() * cl.float().to(DEV)).sum().backward()
    assert rel(high, hr) < tol and rel(x.grad, xr.grad) < 5 * tol
    if halves is None:
        assert float(low.detach().abs().max()) == 0.0
    else:
        assert rel(low, lr) < tol


def test_fre_rect_matches_reference_arithmetic():
    from image_restoration_amd import ops
    sd = R.make_state(A.fre_param_shapes(32, 2), 77, torch.float32)
    feat = seeded_input((4, 32, 384, 256), 78) * 3.0
    ref = A.mask_half_sizes(feat, sd)
    pooled = feat.mean((2, 3)).to(DEV)
    got = ops.fre_rect(pooled, sd["rate_conv.0.weight"].reshape(4, 32).to(DEV), sd["rate_conv.2.weight"].reshape(2, 4).to(DEV), 384, 256)
    assert np.array_equal(got.cpu().numpy(), ref.numpy().astype(np.int32))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_gate_kernels_vs_torch_autograd(dtype, tol):
    """SpatialGate planes, ChannelGate, FreRefine's mix and the para1 / para2 scale-add against plain torch on the CPU (fp64)."""
    from image_restoration_amd import adair
    B, C, H, W = 2, 32, 16, 24
    low, high = seeded_input((B, C, H, W), 40).to(dtype), seeded_input((B, C, H, W), 41).to(dtype)
    sd = R.make_state({"SpatialGate.spatial.weight": (1, 2, 7, 7), "ChannelGate.mlp.0.weight": (2, C, 1, 1),
                       "ChannelGate.mlp.2.weight": (C, 2, 1, 1), "proj.weight": (C, C, 1, 1), "proj.bias": (C,)}, 42)
    lr, hr = low.double().requires_grad_(True), high.double().requires_grad_(True)
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = A.fre_refine(lr, hr, ps)
    cot = seeded_input(tuple(ref.shape), 43)
    ref.backward(cot.double())
    m = adair.FreRefine(C).to(DEV)
    m.load_state_dict(sd)
    lg, hg = low.to(DEV).requires_grad_(True), high.to(DEV).requires_grad_(True)
    out = m(lg, hg)
    out.backward(cot.to(DEV).to(dtype))
    assert rel(out, ref) < max(tol, 1e-4)
    assert rel(lg.grad, lr.grad) < 10 * max(tol, 1e-4) and rel(hg.grad, hr.grad) < 10 * max(tol, 1e-4)
    for k, p in m.named_parameters():
        assert rel(p.grad, ps[k].grad) < 10 * max(tol, 1e-4), k
    # scale-add
    p1, p2 = seeded_input((C, 1, 1), 44), seeded_input((C, 1, 1), 45)
    a64, y64, q1, q2 = lr.detach().requires_grad_(True), hr.detach().requires_grad_(True), p1.double().requires_grad_(True), p2.double().requires_grad_(True)
    (a64 * q1 + y64 * q2).backward(cot.double())
    ag, yg = low.to(DEV).requires_grad_(True), high.to(DEV).requires_grad_(True)
    g1, g2 = p1.to(DEV).requires_grad_(True), p2.to(DEV).requires_grad_(True)
    res = adair._ScaleAddFn.apply(ag, yg, g1, g2)
    res.backward(cot.to(DEV).to(dtype))
    assert rel(res, a64 * q1 + y64 * q2) < max(tol, 1e-5)
    assert rel(ag.grad, a64.grad) < max(tol, 1e-5) and rel(yg.grad, y64.grad) < max(tol, 1e-5)
    assert rel(g1.grad, q1.grad) < max(tol, 1e-4) and rel(g2.grad, q2.grad) < max(tol, 1e-4)


@pytest.mark.parametrize("tag,dim,heads,img_hw,hw,B", [("c32", 32, 2, 64, 16, 2), ("c16_big", 16, 2, 768, 384, 2)])
def test_fre_module_vs_reference_golden(tag, dim, heads, img_hw, hw, B):
    """FreModule forward, dy and every parameter gradient against the reference's own numbers (fp32 activations)."""
    from image_restoration_amd import adair
    gold = load(f"adair_fre_{tag}")
    sd = R.make_state(A.fre_param_shapes(dim, heads), 900 + dim, torch.float32)
    m = adair.FreModule(dim, heads, False).to(DEV)
    m.load_state_dict(sd)
    img = seeded_input((B, 3, img_hw, img_hw), 910 + dim).to(DEV)
    y = seeded_input((B, dim, hw, hw), 911 + dim).to(DEV).requires_grad_(True)
    out = m(img, y)
    out.backward(seeded_input(tuple(out.shape), 912 + dim).to(DEV))
    check("y", out, gold, 2e-4)
    check("dy", y.grad, gold, 2e-3)
    for k, p in m.named_parameters():
        if f"g_{k}.sub" in gold and p.grad is not None:
            check("g_" + k, p.grad, gold, 2e-3, what=k + " ")
    assert m.conv.weight.grad is None and m.rate_conv[0].weight.grad is None     # untouched / non-differentiable, as in the reference


def test_fre_module_bf16_vs_oracle():
    from image_restoration_amd import adair
    dim, heads = 48, 2
    sd = R.make_state(A.fre_param_shapes(dim, heads), 61, torch.float32)
    m = adair.FreModule(dim, heads, False).to(DEV)
    m.load_state_dict(sd)
    img, y0 = seeded_input((2, 3, 128, 128), 62), seeded_input((2, dim, 32, 32), 63)
    cot = seeded_input((2, dim, 32, 32), 64)
    yr = y0.double().requires_grad_(True)
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ref = A.fre_module(img.double(), yr, ps, heads)
    ref.backward(cot.double())
    y = y0.to(DEV).to(torch.bfloat16).requires_grad_(True)
    out = m(img.to(DEV).to(torch.bfloat16), y)
    out.backward(cot.to(DEV).to(torch.bfloat16))
    assert rel(out, ref) < 3e-2 and rel(y.grad, yr.grad) < 3e-2
    for k, p in m.named_parameters():
        if p.grad is not None and ps[k].grad is not None:
            # conv1 feeds |.|: where the bf16 conv output lands on the other side of zero the gradient's sign flips (a kink, not a
            # rounding error); ~1 % of the pixels at these magnitudes -> 15 % on the 3 -> 48 conv's weight gradient
            assert rel(p.grad, ps[k].grad) < (0.15 if k == "conv1.weight" else 4e-2), k


def test_adair_tiny_network_vs_reference_golden():
    """The assembled AdaIR network (tiny widths): train-mode output, L1 loss and every parameter-gradient norm against the
    reference (fp32 activations); state_dict keys of the base configuration equal the reference's."""
    from image_restoration_amd import adair
    cfg = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 2], ffn_expansion_factor=2.66,
               bias=False, LayerNorm_type="WithBias", decoder=True)
    gold = load("adair_tiny_train")
    net = adair.AdaIR(**cfg).to(DEV)
    net.load_state_dict(R.make_state(A.adair_param_shapes(cfg), 950, torch.float32))
    img = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(951), dtype=F64).float().to(DEV)
    tgt = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(952), dtype=F64).float().to(DEV)
    out = net(img)
    loss = (out - tgt).abs().mean()
    loss.backward()
    check("y", out, gold, 5e-4)
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 * max(1.0, float(gold["loss"]))
    grads = dict(net.named_parameters())
    for k, n in zip(gold["grad_names"], gold["grad_norms"]):
        p = grads[str(k)]
        if n < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        elif n > 0:
            assert abs(float(p.grad.norm()) - n) <= 3e-3 * n + 1e-7, (str(k), float(p.grad.norm()), n)
    keys = load("adair_keys")
    assert list(adair.AdaIR().state_dict()) == [str(k) for k in keys["adair_base"]]
