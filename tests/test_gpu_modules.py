"""GPU parity of the drop-in modules (reference interface) against the golden vectors captured from the
reference and against the fp64 CPU oracle: forward, input gradient and every parameter gradient.
Tolerances: fp32 activations 1e-3 relative (north_star bar; observed ~1e-5), bf16 looser and stated per test."""
import numpy as np
import pytest
import torch

from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input

pytestmark = pytest.mark.gpu
DEV = "cuda"


def M():
    import image_restoration_amd as m
    return m


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def run(mod, x, cot, dtype=torch.float32):
    mod = mod.to(DEV)
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = mod(xg)
    y.backward(cot.to(DEV).to(dtype))
    return y, xg.grad, {k: p.grad for k, p in mod.named_parameters()}


CASES = [("c48h1", 48, 1, (2, 48, 16, 16), False), ("c48h1_bias", 48, 1, (2, 48, 16, 16), True),
         ("c16h1", 16, 1, (2, 16, 16, 16), False), ("c96h2", 96, 2, (2, 96, 16, 16), False),
         ("c96h1", 96, 1, (2, 96, 16, 16), False), ("c48h1_64", 48, 1, (1, 48, 64, 64), False)]


@pytest.mark.parametrize("tag,c,heads,shape,bias", CASES)
def test_modules_vs_reference_golden(tag, c, heads, shape, bias):
    """Same seeded parameters and inputs as tools/capture_golden.py; expected values come from the reference."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, bias, "WithBias", seed=7 + c + heads)
    x = seeded_input(shape, 200 + c + heads)
    for name, mod, sub, seed in (("ffn", m.FeedForward(c, 2.66, bias), "ffn.", 300),
                                 ("attn", m.Attention(c, heads, bias), "attn.", 400),
                                 ("block", m.TransformerBlock(c, heads, 2.66, bias, "WithBias"), "", 500)):
        mod.load_state_dict(R.sub_state(sd, sub) if sub else sd)
        cot = seeded_input(shape, seed + 1000)
        y, dx, g = run(mod, x, cot)
        gold = load(f"{name}_{tag}")
        check("y", y, gold, 1e-3, what=name + " ")
        check("dx", dx, gold, 1e-3, what=name + " ")
        for k, v in g.items():
            check("g_" + k, v, gold, 1e-3, what=name + " ")


def test_block_biasfree_vs_reference_golden():
    m = M()
    sd = R.make_block_state(48, 1, 2.66, False, "BiasFree", seed=77)
    blk = m.TransformerBlock(48, 1, 2.66, False, "BiasFree")
    blk.load_state_dict(sd)
    x = seeded_input((2, 48, 16, 16), 277)
    y, dx, g = run(blk, x, seeded_input((2, 48, 16, 16), 1500))
    gold = load("block_c48h1_biasfree")
    check("y", y, gold, 1e-3)
    check("dx", dx, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3)


@pytest.mark.parametrize("c,heads,shape", [(192, 4, (1, 192, 8, 8)), (384, 8, (2, 384, 4, 4)), (48, 1, (1, 48, 9, 11)),
                                           (32, 2, (3, 32, 5, 16))])
def test_block_vs_oracle_tight(c, heads, shape):
    """Deeper-level shapes (C=192/384, 4/8 heads) and ragged H,W against the fp64 oracle, tight bound 5e-5."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, True, "WithBias", seed=c + heads)
    blk = m.TransformerBlock(c, heads, 2.66, True, "WithBias")
    blk.load_state_dict(sd)
    x = seeded_input(shape, 900 + c)
    cot = seeded_input(shape, 901 + c)
    y, dx, g = run(blk, x, cot)
    xr = x.double().requires_grad_(True)
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, sdr, heads, "WithBias")
    yr.backward(cot.double())
    assert rel(y, yr) < 5e-5
    assert rel(dx, xr.grad) < 5e-5
    for k, v in g.items():
        assert rel(v, sdr[k].grad) < 5e-5, k


def test_block_bf16_activations():
    """bf16 storage / bf16 MFMA with fp32 accumulation: 3e-2 relative on outputs and gradients."""
    m = M()
    c, heads, shape = 48, 1, (2, 48, 32, 32)
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=5)
    blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias")
    blk.load_state_dict(sd)
    x = seeded_input(shape, 910).bfloat16().float()
    cot = seeded_input(shape, 911).bfloat16().float()
    y, dx, g = run(blk, x, cot, torch.bfloat16)
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16
    xr = x.double().requires_grad_(True)
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, sdr, heads, "WithBias")
    yr.backward(cot.double())
    assert rel(y, yr) < 3e-2
    assert rel(dx, xr.grad) < 3e-2
    for k, v in g.items():
        assert v.dtype == torch.float32
        assert rel(v, sdr[k].grad) < 3e-2, k


def test_main_grad_accumulation_and_inference_mode():
    m = M()
    c, heads, shape = 48, 1, (1, 48, 16, 16)
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=6)
    blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias")
    blk.load_state_dict(sd)
    x, cot = seeded_input(shape, 920), seeded_input(shape, 921)
    _, _, g = run(blk, x, cot)
    blk2 = m.TransformerBlock(c, heads, 2.66, False, "WithBias")
    blk2.load_state_dict(sd)
    blk2 = blk2.to(DEV)
    for p in blk2.parameters():
        p.main_grad = torch.zeros_like(p)
    for _ in range(2):
        xg = x.to(DEV).requires_grad_(True)
        blk2(xg).backward(cot.to(DEV))
    for (k, p) in blk2.named_parameters():
        assert p.grad is None
        assert rel(p.main_grad, 2 * g[k].cpu()) < 1e-5, k
    with torch.no_grad():
        y0 = blk2(x.to(DEV))
    assert rel(y0, blk.to(DEV)(x.to(DEV))) < 1e-6


def test_restormer_tiny_config_c1_and_psnr():
    """BASELINE config 1 on the GPU: Restormer-tiny, sigma=25, 1x3x128x128; output within 1e-3 of the reference's
    fp64 forward, PSNR within 0.01 dB, loss-gradient norms within 1e-3."""
    m = M()
    gold = load("restormer_tiny_128")
    cfg = R.RESTORMER_TINY
    net = m.Restormer(**cfg)
    net.load_state_dict(R.make_restormer_state(cfg, seed=1))
    net = net.to(DEV)
    clean = torch.from_numpy(np.random.default_rng(1234).random((1, 3, 128, 128))).to(torch.float32)
    noisy = R.degrade_sigma(clean, 25.0, seed=4321)
    out = net(noisy.to(DEV))
    check("y64", out[:, :, ::4, ::4], gold, 1e-3)
    assert abs(R.psnr(out.cpu(), clean) - float(gold["psnr_out"])) < 0.01
    loss = (out - clean.to(DEV)).abs().mean()
    assert abs(loss.item() - float(gold["loss"])) < 1e-4 * float(gold["loss"])
    loss.backward()
    keys = [str(k) for k in gold["grad_norm_keys"]]
    ref = np.asarray(gold["grad_norms"])
    params = dict(net.named_parameters())
    got = np.array([float(params[k].grad.norm()) for k in keys])
    assert np.all(np.abs(got - ref) <= 1e-3 * np.maximum(ref, 1e-12) + 1e-9), np.abs(got / ref - 1).max()


def test_packed_weight_cache_is_transparent():
    """mi_pw_cache_*: with the cache on, results are bit-identical to per-call packing, before and after an in-place
    weight update followed by refresh; a stale cache is avoided by invalidate."""
    from image_restoration_amd import ops
    m = M()
    c, heads, shape = 48, 1, (2, 48, 16, 16)
    sd = R.make_block_state(c, heads, 2.66, True, "WithBias", seed=8)
    x, cot = seeded_input(shape, 930), seeded_input(shape, 931)

    def fresh(scale):
        blk = m.TransformerBlock(c, heads, 2.66, True, "WithBias")
        blk.load_state_dict(sd)
        blk = blk.to(DEV)
        with torch.no_grad():
            for p in blk.parameters():
                p.mul_(scale)
        return blk

    def go(blk):
        for p in blk.parameters():
            p.grad = None
        y, dx, g = run(blk, x, cot)
        return [y, dx] + [g[k] for k in sorted(g)]

    ref1, ref2 = go(fresh(1.0)), go(fresh(2.0))   # powers of two: the in-place updates below are exact
    try:
        blk = fresh(1.0)
        # the cache only takes weights inside the registered parameter storage: move the block's parameters into one
        flat = torch.cat([p.detach().reshape(-1) for p in blk.parameters()]).contiguous()
        off = 0
        for p in blk.parameters():
            p.data = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        ops.pw_cache_enable(32 << 20, torch.device(DEV), flat)
        go(blk)                      # registers the block's weights (still packs per call)
        ops.pw_cache_refresh()
        for a, b in zip(go(blk), ref1):
            assert torch.equal(a, b)
        with torch.no_grad():
            for p in blk.parameters():
                p.mul_(2.0)
        ops.pw_cache_refresh()       # what the trainer does after the optimizer step
        for a, b in zip(go(blk), ref2):
            assert torch.equal(a, b)
        with torch.no_grad():
            for p in blk.parameters():
                p.mul_(0.5)
        ops.pw_cache_invalidate()    # weights changed without a refresh: must not use the stale images
        for a, b in zip(go(blk), ref1):
            assert torch.equal(a, b)
    finally:
        ops.pw_cache_enable(0, torch.device(DEV), None)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("cin,cout,hw,bias", [(3, 48, (40, 64), False), (96, 3, (24, 128), False), (16, 3, (9, 32), True),
                                              (3, 16, (33, 256), True)])
def test_thin_glue_conv3x3_native(dtype, tol, cin, cout, hw, bias):
    """OverlapPatchEmbed (3 -> dim) and the output conv (2*dim -> 3, + input residual) through the native
    im2col / col2im + 1x1 GEMM / Gram path (Restormer.py:156-165,243,281) against F.conv2d in fp64."""
    import torch.nn as nn
    import torch.nn.functional as F
    from image_restoration_amd import restormer as RM
    torch.manual_seed(5)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
    H, W = hw
    x = seeded_input((2, cin, H, W), 950)
    res = seeded_input((2, cout, H, W), 951)
    cot = seeded_input((2, cout, H, W), 952)
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    br = conv.bias.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, 1, 1) + res.double()
    yr.backward(cot.double())
    conv = conv.to(DEV)
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    assert isinstance(RM._conv2d(xg, conv, res.to(DEV).to(dtype)).grad_fn, RM._Conv3x3Fn._backward_cls)
    y = RM._conv2d(xg, conv, res.to(DEV).to(dtype))
    y.backward(cot.to(DEV).to(dtype))
    assert rel(y, yr) < tol
    assert rel(xg.grad, xr.grad) < tol
    assert rel(conv.weight.grad, wr.grad) < tol
    if bias:
        assert rel(conv.bias.grad, br.grad) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("c,heads,hw", [(48, 1, (256, 256)), (96, 1, (256, 256)), (96, 2, (128, 128)), (192, 4, (64, 64)),
                                        (384, 8, (32, 32))])
def test_block_at_training_resolution_vs_oracle(dtype, tol, c, heads, hw):
    """Every level of Restormer base at its REAL plane size for 256x256 training (BASELINE configs 1-2: level 1 and
    decoder/refinement at 256^2, level 2 at 128^2, level 3 at 64^2, latent at 32^2), two images, forward and all gradients
    against the oracle (fp32 on the host: fp64 at these sizes would take minutes).  This is where the full-width streaming
    depthwise kernels (64 lanes per row), the weight-resident GEMM, the streaming Gram with many pixel splits and the
    wave-owned LayerNorms actually run; the small-shape tests above cannot reach them."""
    m = M()
    H, W = hw
    shape = (2, c, H, W)
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=c + 7 * heads)
    blk = m.TransformerBlock(c, heads, 2.66, False, "WithBias")
    blk.load_state_dict(sd)
    x = seeded_input(shape, 960 + c)
    cot = seeded_input(shape, 961 + c)
    if dtype == torch.bfloat16:
        x, cot = x.bfloat16().float(), cot.bfloat16().float()
    y, dx, g = run(blk, x, cot, dtype)
    torch.set_num_threads(min(32, torch.get_num_threads() or 1) or 1)
    xr = x.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, sdr, heads, "WithBias")
    yr.backward(cot)
    assert rel(y, yr) < tol
    assert rel(dx, xr.grad) < tol
    for k, v in g.items():
        # the temperature gradient of a normalised-logit softmax is a small difference of large sums over 65536 pixels:
        # fp32 host arithmetic itself is only good to ~1e-3 there
        bound = 10 * tol if k.endswith("temperature") else tol
        assert rel(v, sdr[k].grad) < bound, k


@pytest.mark.parametrize("env", ["MI_GRAM_LDS", "MI_GRAM_STREAM_ALL", "MI_PW_CHUNKED", "MI_PW_TM_EVEN", "MI_DW_LDS",
                                 "MI_GDFN_STORE_Y", "MI_LN_FORM=block", "MI_LN_FORM=wave", "MI_PW_DMA", "MI_PW_WAVE=0"])
def test_alternate_kernel_paths_stay_correct(monkeypatch, env):
    """Every A/B switch selects a kernel that is otherwise only reached for other shapes (or not at all): run one block
    at a training-size plane through each and hold it to the same bound as the default path."""
    name, _, val = env.partition("=")
    monkeypatch.setenv(name, val or "1")
    m = M()
    c, heads, shape = 96, 2, (2, 96, 64, 128)
    sd = R.make_block_state(c, heads, 2.66, True, "WithBias", seed=77)
    blk = m.TransformerBlock(c, heads, 2.66, True, "WithBias")
    blk.load_state_dict(sd)
    x, cot = seeded_input(shape, 970), seeded_input(shape, 971)
    y, dx, g = run(blk, x, cot)
    xr = x.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, sdr, heads, "WithBias")
    yr.backward(cot)
    assert rel(y, yr) < 2e-4
    assert rel(dx, xr.grad) < 2e-4
    for k, v in g.items():
        assert rel(v, sdr[k].grad) < (2e-3 if k.endswith("temperature") else 2e-4), k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("kind,c,hw", [("down", 48, (32, 64)), ("down", 96, (16, 32)), ("up", 96, (16, 32)), ("up", 64, (24, 16))])
def test_down_up_sample_native(dtype, tol, kind, c, hw):
    """Downsample (3x3 C -> C/2 + PixelUnshuffle) and Upsample (3x3 C -> 2C + PixelShuffle) (Restormer.py:171-189) through the
    native im2col / col2im GEMM forms and the native pixel (un)shuffle, against torch in fp64: output, input and weight grads."""
    import torch.nn.functional as F
    m = M()
    torch.manual_seed(11)
    mod = (m.Downsample(c) if kind == "down" else m.Upsample(c))
    H, W = hw
    x = seeded_input((2, c, H, W), 960)
    wr = mod.body[0].weight.detach().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr = F.pixel_unshuffle(yr, 2) if kind == "down" else F.pixel_shuffle(yr, 2)
    cot = seeded_input(tuple(yr.shape), 961)
    yr.backward(cot.double())
    mod = mod.to(DEV)
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = mod(xg)
    assert "PixelShuffle" in type(y.grad_fn).__name__, "native pixel (un)shuffle not taken"
    y.backward(cot.to(DEV).to(dtype))
    assert rel(y, yr) < tol and rel(xg.grad, xr.grad) < tol and rel(mod.body[0].weight.grad, wr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_up_cat_writes_into_concat_buffer(dtype):
    """cat([PixelShuffle(z), skip]) without the intermediate (decoder level 1, Restormer.py:265-266): exact data movement."""
    import torch.nn.functional as F
    from image_restoration_amd import restormer as RM
    z = seeded_input((2, 64, 8, 16), 970).to(DEV).to(dtype).requires_grad_(True)
    skip = seeded_input((2, 16, 16, 32), 971).to(DEV).to(dtype).requires_grad_(True)
    out = RM._UpCatFn.apply(z, skip)
    ref = torch.cat([F.pixel_shuffle(z.detach(), 2), skip.detach()], 1)
    assert torch.equal(out, ref)
    cot = seeded_input(tuple(out.shape), 972).to(DEV).to(dtype)
    out.backward(cot)
    assert torch.equal(z.grad, F.pixel_unshuffle(cot[:, :16], 2)) and torch.equal(skip.grad, cot[:, 16:])


@pytest.mark.parametrize("c,heads,shape", [(48, 1, (2, 48, 32, 64)), (96, 2, (1, 96, 16, 32))])
def test_block_fp32_elementwise_relative(c, heads, shape):
    """Every other bound in this file is max|delta| / max|ref| (one scale per tensor).  Here the fp32 path is also held to an
    ELEMENT-WISE bound against the fp64 oracle - |delta_i| <= 1e-3 |ref_i| + 1e-5 max|ref| for every element of the output, the
    input gradient and every parameter gradient - which is the literal reading of north_star's "1e-3 relative fp32"."""
    m = M()
    sd = R.make_block_state(c, heads, 2.66, True, "WithBias", seed=5 + c)
    blk = m.TransformerBlock(c, heads, 2.66, True, "WithBias")
    blk.load_state_dict(sd)
    x, cot = seeded_input(shape, 6100 + c), seeded_input(shape, 6101 + c)
    y, dx, g = run(blk, x, cot)
    xr = x.double().requires_grad_(True)
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, ps, heads, "WithBias")
    yr.backward(cot.double())

    def elementwise(got, ref, what):
        ref = ref.detach().double()
        d = (got.detach().cpu().double() - ref).abs()
        bound = 1e-3 * ref.abs() + 1e-5 * ref.abs().max()
        bad = int((d > bound).sum())
        assert bad == 0, f"{what}: {bad} of {d.numel()} elements outside 1e-3 relative (+1e-5 of the tensor's max); worst {float((d / bound).max()):.2f}x"
    elementwise(y, yr, "y")
    elementwise(dx, xr.grad, "dx")
    for k, v in g.items():
        elementwise(v, ps[k].grad, "g_" + k)


@pytest.mark.parametrize("name,dtype,ltol,ptol", [("tiny", torch.float32, 1e-4, None), ("dim48", torch.bfloat16, 2e-2, None)])
def test_training_steps_follow_the_oracle_trajectory(name, dtype, ltol, ptol):
    """Three whole training steps (forward, L1, backward, AdamW) on the GPU through FlatTrainer - native kernels, main_grad
    accumulation into the flat buffer, the fused AdamW kernel - against the CPU oracle driven by torch autograd and
    torch.optim.AdamW (MoCE-IR-main/src/train.py:50-88: L1Loss + AdamW): the loss of every step and every parameter after the
    last step.  fp32 activations (Restormer-tiny): every loss to 1e-4; parameters in AdamW's own units (below).  bf16 activations at the real block widths
    (dim 48, one block per level: the LayerNorm-in-GEMM head and the one-launch backward tails run at C = 48 and 96): losses to 2e-2,
    parameters in units of the learning rate - the first AdamW steps move every weight by ~lr whatever the gradient's size, so a
    bf16-flipped sign of a tiny gradient shows up as 2 lr per step on that weight (bound: 6.3 lr worst case, 0.4 lr on average; measured up to 5.8 and 0.27)."""
    m = M()
    from image_restoration_amd.trainer import FlatTrainer
    cfg = R.RESTORMER_TINY if name == "tiny" else R.restormer_config(dim=48, num_blocks=(1, 1, 1, 1), num_refinement_blocks=1)
    sd0 = R.make_restormer_state(cfg, seed=2)
    clean = torch.from_numpy(np.random.default_rng(77).random((2, 3, 64, 64))).to(torch.float32)
    noisy = R.degrade_sigma(clean, 25.0, seed=78)
    lr = 1e-3
    # oracle
    ps = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    opt = torch.optim.AdamW(list(ps.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ref_losses = []
    gfloor = {}          # per element: the smallest |gradient| (relative to the tensor's largest) the oracle saw in the three steps
    for _ in range(3):
        opt.zero_grad()
        loss = (R.restormer_forward(noisy, ps, cfg) - clean).abs().mean()
        loss.backward()
        for k, v in ps.items():
            gr = v.grad.detach().abs() / v.grad.detach().abs().max().clamp_min(1e-30)
            gfloor[k] = gr if k not in gfloor else torch.minimum(gfloor[k], gr)
        opt.step()
        ref_losses.append(float(loss.detach()))
    # product
    net = m.Restormer(**cfg)
    net.load_state_dict(sd0)
    net = net.to(DEV).train()
    tr = FlatTrainer(net, lr=lr, weight_decay=0.01)
    try:
        x, y = noisy.to(DEV).to(dtype), clean.to(DEV).to(dtype)
        for step in range(3):
            tr.zero_grad()
            loss = (net(x).float() - y.float()).abs().mean()
            loss.backward()
            tr.reduce_gradients()
            tr.optimizer_step()
            assert abs(float(loss) - ref_losses[step]) < ltol * ref_losses[step], (step, float(loss), ref_losses[step])
        got = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    finally:
        tr.close()
    assert ref_losses[2] < ref_losses[0]                       # the steps do reduce the loss
    for k, v in ps.items():
        if dtype == torch.float32:
            # The first AdamW steps move a weight by lr * g / (|g| + eps) - every element by about +-lr whatever its gradient's size -
            # so an element whose gradient is within the fp32 noise of the two evaluations has its update decided by that noise, and
            # the displaced weights then perturb the later steps' gradients.  torch's own CPU fp32 and fp64 trajectories of this
            # test end up to 0.21 lr apart in single elements (tools/debug_traj3.py), so a bound relative to max|w| (the round-1
            # form of this check: 2e-3, i.e. 0.1 lr on the conv weights) only ever held by luck of the rounding pattern.  Stated
            # in AdamW's own units instead, per tensor: the UPDATE (w - w0) points the same way (cosine >= 0.9995; measured
            # 0.99997), the mean displacement is <= 0.02 lr (measured 0.0065) and no element is more than 2 lr away (measured 0.93
            # - less than one opposite step); elements whose gradient stayed above 1 % of the tensor's largest are held to 0.5 lr
            # (measured 0.204).
            w0 = sd0[k].double()
            a, r = got[k].double(), v.detach().double()
            d = (a - r).abs()
            ua, ur = (a - w0).flatten(), (r - w0).flatten()
            cos = float((ua @ ur) / (ua.norm() * ur.norm()).clamp_min(1e-30))
            assert cos >= 0.9995, (k, cos)
            assert float(d.mean()) <= 0.02 * lr and float(d.max()) <= 2.0 * lr, (k, float(d.mean()) / lr, float(d.max()) / lr)
            big = gfloor[k] > 1e-2
            if bool(big.any()):
                assert float(d[big].max()) <= 0.5 * lr, (k, float(d[big].max()) / lr)
        else:
            # absolute, in units of lr: an AdamW step moves a weight by at most ~lr (bias-corrected m / sqrt(v) <= 1), so two runs
            # that disagree on the sign of a tiny gradient in all three steps end 2 * 3 lr apart; most weights agree far better
            d = (got[k].double() - v.detach().double()).abs()
            assert float(d.max()) < 6.3 * lr and float(d.mean()) < 0.4 * lr, (k, float(d.max()) / lr, float(d.mean()) / lr)
