"""GPU parity of the MoCE-IR / AdaIR drop-in modules against the reference's golden vectors
(tools/capture_golden_moce.py) and the fp64 oracle.  fp32 activations, 1e-3 relative (north_star bar)."""
import numpy as np
import pytest
import torch

from oracle import moce_ref as MR
from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input
from test_oracle_golden_moce import cross_shapes, decoder_state, encoder_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run(mod, inputs, seed, call=None):
    mod = mod.to(DEV)
    ins = [t.to(DEV).requires_grad_(True) for t in inputs]
    y = (call or mod)(*ins)
    y0 = y[0] if isinstance(y, tuple) else y
    y0.backward(seeded_input(tuple(y0.shape), seed + 1000).to(DEV))
    return y, [t.grad for t in ins], {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}


class injected_noise:
    def __init__(self, seed):
        self.seed = seed

    def __enter__(self):
        self.orig = torch.randn_like
        seed = self.seed
        torch.randn_like = lambda t, **kw: seeded_input(tuple(t.shape), seed, torch.float64).to(t.dtype).to(t.device)

    def __exit__(self, *a):
        torch.randn_like = self.orig


@pytest.mark.parametrize("tag,c,heads", [("moce_cross_c48h1", 48, 1), ("moce_cross_c96h2", 96, 2)])
def test_cross_attention_moce(tag, c, heads):
    import image_restoration_amd.moce_ir as mo
    m = mo.CrossAttention(c, heads, True)
    m.load_state_dict(R.make_state(cross_shapes(c, heads, True, 7), 60 + c))
    x, y = seeded_input((2, c, 16, 16), 600 + c), seeded_input((2, c, 16, 16), 601 + c)
    out, (dx, dy), g = run(m, [x, y], 610)
    gold = load(tag)
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3); check("dy", dy, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3)


def test_cross_attention_adair():
    import image_restoration_amd.adair as ad
    m = ad.Chanel_Cross_Attention(48, 4, False)
    m.load_state_dict(R.make_state(cross_shapes(48, 4, False, 3), 65))
    x, y = seeded_input((2, 48, 16, 16), 650), seeded_input((2, 48, 16, 16), 651)
    out, (dx, dy), g = run(m, [x, y], 660)
    gold = load("adair_cross_c48h4")
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3); check("dy", dy, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3)
    with pytest.raises(AssertionError):
        m(x.to(DEV), y[:, :, :8].contiguous().to(DEV))


def test_encoder_block():
    import image_restoration_amd.moce_ir as mo
    m = mo.EncoderBlock(48, 2, 2, True, "WithBias")
    m.load_state_dict(R.make_state(encoder_shapes(48, 2, 2, True), 75))
    out, (dx,), g = run(m, [seeded_input((2, 48, 16, 16), 750)], 760)
    gold = load("moce_encoder_c48h2")
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3)


@pytest.mark.parametrize("k", [1, 2])
def test_routing_and_dispatcher(k):
    import image_restoration_amd.moce_ir as mo
    gold = load(f"moce_routing_k{k}")
    comp = torch.tensor([18840., 42288., 103008., 279744.])
    rf = mo.RoutingFunction(48, 64, num_experts=4, k=k, complexity=comp, use_complexity_bias=True, complexity_scale="max")
    rf.load_state_dict(R.make_state({"gate.2.weight": (4, 48), "freq_gate.weight": (4, 64)}, 70 + k), strict=False)
    rf = rf.to(DEV).train()
    x, fe = seeded_input((8, 48, 8, 8), 700).to(DEV), seeded_input((8, 64), 701).to(DEV)
    with injected_noise(702):
        gates, idx, vals, aux = rf(x, fe)
    check("gates", gates, gold, 1e-4)
    assert np.array_equal(idx.cpu().numpy(), gold["idx"])
    assert abs(float(aux) - float(gold["aux"])) < 1e-4
    disp = mo.SparseDispatcher(4, gates)
    parts = disp.dispatch(x)
    assert [p.shape[0] for p in parts] == list(gold["part_sizes"])
    comb = disp.combine([p * (e + 1) for e, p in enumerate(parts)], multiply_by_gates=True)
    assert comb.dtype == torch.float32
    check("combined", comb, gold, 1e-4)


def test_decoder_block_train_and_eval():
    gold = load("moce_decoder_train")
    m, sd = decoder_state(torch.float32)
    m.load_state_dict(sd, strict=False)
    assert np.allclose(m.adapter.routing.complexity.numpy(), gold["complexity"])
    x, fe = seeded_input((4, 48, 16, 16), 800), seeded_input((4, 64), 801)
    m.train()
    with injected_noise(802):
        (out, aux), (dx, dfe), g = run(m, [x, fe], 810)
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3)
    assert abs(float(aux) - float(gold["aux"])) < 1e-4
    for k, v in g.items():
        if "g_" + k + ".l2" not in gold.files:
            # an expert no sample was routed to: the reference leaves its .grad None; so does the ragged dispatch - the capacity
            # dispatch (no host read-back of the segment sizes) runs it over zero rows and reports an exactly zero gradient
            assert float(v.abs().max()) == 0.0, k
            continue
        if float(gold["g_" + k + ".l2"]) < 1e-6:      # router weights: no gradient through the main path at top_k = 1
            assert float(v.norm()) < 1e-5, k
            continue
        check("g_" + k, v, gold, 2e-3, what="decoder ")
    m.eval()
    gold_e = load("moce_decoder_eval")
    with torch.no_grad(), injected_noise(803):
        out_e, aux_e = m(x[:1].to(DEV), fe[:1].to(DEV))
    check("y", out_e, gold_e, 1e-3)
    assert aux_e == 0


def test_decoder_block_bf16_runs():
    """bf16 activations through the whole MoCE decoder block (experts' FFT part runs in fp32 as in the reference)."""
    m, sd = decoder_state(torch.float32)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).train()
    x = seeded_input((4, 48, 16, 16), 800).to(DEV).bfloat16().requires_grad_(True)
    fe = seeded_input((4, 64), 801).to(DEV)
    with injected_noise(802):
        out, aux = m(x, fe)
    assert out.dtype == torch.bfloat16
    out.float().mean().backward()
    ref, _ = MR.decoder_block(x.detach().float().cpu().double(), fe.cpu().double(), {k: v.double() for k, v in sd.items()}, 1,
                              dict(dim=48, rank=2, num_experts=4, top_k=1, rank_type="spread", with_complexity=True,
                                   complexity=m.adapter.routing.complexity.cpu().double()),
                              seeded_input((4, 4), 802, torch.float64), True)
    err = float((out.float().cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-2, err


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dispatcher_native_rows_forward_and_backward(dtype):
    """SparseDispatcher.dispatch / combine through the native row kernels against the reference's torch formulation
    (index gather, gate multiply, index_add into fp32 zeros: moce_ir.py:103-124), including the gradients w.r.t. the
    dispatched input, the expert outputs and the gate values; top-2 routing so that every sample is hit twice."""
    import image_restoration_amd.moce_ir as mo
    torch.manual_seed(3)
    B, E, shape = 6, 4, (6, 12, 9, 16)
    logits = torch.randn(B, E)
    top = logits.topk(2, dim=1)
    gates = torch.zeros(B, E).scatter_(1, top.indices, torch.softmax(top.values, 1)).to(DEV).requires_grad_(True)
    x = seeded_input(shape, 980).to(DEV).to(dtype).requires_grad_(True)
    cot = seeded_input(shape, 981).to(DEV)
    # native path
    disp = mo.SparseDispatcher(E, gates)
    parts = disp.dispatch(x)
    out = disp.combine([p * (e + 1.5) for e, p in enumerate(parts)], multiply_by_gates=True)
    assert out.dtype == torch.float32
    out.backward(cot)
    gx, gg = x.grad.clone(), gates.grad.clone()
    # reference formulation in fp64 on the same routing
    x.grad = None; gates.grad = None
    xr = x.detach().double().requires_grad_(True)
    gr = gates.detach().double().requires_grad_(True)
    bi, nzg = disp._batch_index.flatten(), None
    gates_exp = gr[bi]
    nzg = torch.gather(gates_exp, 1, disp._expert_index)
    inp_exp = xr[bi]
    sizes = disp._part_sizes
    outs = [p * (e + 1.5) for e, p in enumerate(torch.split(inp_exp, sizes, 0))]
    st = torch.cat(outs, 0) * nzg.unsqueeze(-1).unsqueeze(-1)
    ref = torch.zeros(B, *shape[1:], dtype=torch.float64, device=DEV).index_add(0, bi, st)
    ref.backward(cot.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(out, ref) < tol
    assert rel(gx, xr.grad) < tol
    assert rel(gg, gr.grad) < tol


# ------------------------------------------------------------------------------------------------ round 2: stand-alone pieces
def _filled(mod, seed):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if not k.endswith("complexity")}
    mod.load_state_dict(R.make_state(shapes, seed), strict=False)
    return mod


@pytest.mark.parametrize("tag,r,p,shape", [("r12p8", 12, 8, (2, 12, 16, 16)), ("r24p16", 24, 16, (1, 24, 20, 12))])
def test_fft_attention_standalone(tag, r, p, shape):
    """FFTAttention alone (native patch circular convolution, ragged plane = the zero-padding path) vs the reference."""
    import image_restoration_amd.moce_ir as mo
    m = _filled(mo.FFTAttention(r, patch_size=p, kernel_size=3), 90 + r)
    out, (dx,), g = run(m, [seeded_input(shape, 900 + r)], 910 + r)
    gold = load(f"moce_fftattn_{tag}")
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3)
    assert len(g) == len(list(m.parameters()))
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3, what="fftattn ")


def test_mod_expert_standalone():
    import image_restoration_amd.moce_ir as mo
    m = _filled(mo.ModExpert(48, rank=12, func=mo.FFTAttention, depth=1, patch_size=8, kernel_size=5), 95)
    out, (dx, dsh), g = run(m, [seeded_input((2, 48, 16, 16), 950), seeded_input((2, 48, 16, 16), 951)], 960)
    gold = load("moce_modexpert_c48r12")
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3); check("dshared", dsh, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3, what="modexpert ")


@pytest.mark.parametrize("k", [1, 2])
def test_adapter_layer_all_experts(k):
    """AdapterLayer alone with a routing that uses all four experts (top-1) / hits every sample twice (top-2): every
    expert's parameters receive the reference's gradient."""
    import image_restoration_amd.moce_ir as mo
    gold = load(f"moce_adapter_k{k}")
    m = _filled(mo.AdapterLayer(48, rank=2, num_experts=4, top_k=k, expert_layer=mo.FFTAttention, stage_depth=1,
                                depth_type="constant", rank_type="spread", freq_dim=64, with_complexity=True,
                                complexity_scale="max"), 100 + k).train()
    ins = [seeded_input((8, 48, 16, 16), 1000), seeded_input((8, 64), 1001), seeded_input((8, 48, 16, 16), 1002)]
    with injected_noise(int(gold["noise_seed"])):
        out, (dx, dfe, dsh), g = run(m, ins, 1010 + k)
    assert np.array_equal(m.routing.tables.perm_expert.cpu().numpy().astype(np.int64).tolist(),
                          sorted(int(e) for e in gold["idx"].reshape(-1)))
    if k == 1:
        assert sorted(set(int(e) for e in gold["idx"].reshape(-1))) == [0, 1, 2, 3]
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3); check("dshared", dsh, gold, 1e-3)
    check("dfe", dfe, gold, 2e-3)
    assert abs(float(m.loss) - float(gold["aux"])) < 1e-4
    names = [n for n, _ in m.named_parameters()]
    for name in names:
        key = "g_" + name + ".l2"
        if key not in gold.files:
            continue
        if float(gold[key]) < 1e-7:
            assert name not in g or float(g[name].norm()) < 1e-5, name
        else:
            check("g_" + name, g[name], gold, 2e-3, what="adapter ")


@pytest.mark.parametrize("k", [1, 2])
def test_router_kernel_gradients(k):
    """mi_moe_route_bwd: gradients of sum(gates * cot) + aux w.r.t. the feature map, the frequency embedding and both gate
    matrices against the reference's autograd (fp64)."""
    import image_restoration_amd.moce_ir as mo
    gold = load(f"moce_router_grads_k{k}")
    comp = torch.tensor([18840., 42288., 103008., 279744.])
    rf = mo.RoutingFunction(48, 64, num_experts=4, k=k, complexity=comp, use_complexity_bias=True, complexity_scale="max")
    rf.load_state_dict(R.make_state({"gate.2.weight": (4, 48), "freq_gate.weight": (4, 64)}, 70 + k), strict=False)
    rf = rf.to(DEV).train()
    x = seeded_input((8, 48, 8, 8), 700).to(DEV).requires_grad_(True)
    fe = seeded_input((8, 64), 701).to(DEV).requires_grad_(True)
    with injected_noise(702):
        gates, idx, vals, aux = rf(x, fe)
    ((gates * seeded_input((8, 4), 703).to(DEV)).sum() + aux).backward()
    check("gates", gates, gold, 1e-4)
    assert abs(float(aux) - float(gold["aux"])) < 1e-4
    check("dx", x.grad, gold, 1e-3); check("dfe", fe.grad, gold, 1e-3)
    check("g_gate", rf.gate[2].weight.grad, gold, 1e-3); check("g_freq", rf.freq_gate.weight.grad, gold, 1e-3)


def test_frequency_embedding():
    import image_restoration_amd.moce_ir as mo
    gold = load("moce_freqemb_d64")
    m = mo.FrequencyEmbedding(64)
    # the fixed Laplacian initialisation before any weights are loaded (moce_ir.py:241-243)
    assert np.allclose(m.high_conv[0].conv.weight[5, 0].detach().numpy(), [[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]])
    m = _filled(m, 110)
    out, (dx,), g = run(m, [seeded_input((2, 64, 8, 8), 1100)], 1110)
    check("y", out, gold, 1e-3); check("dx", dx, gold, 1e-3)
    for k, v in g.items():
        check("g_" + k, v, gold, 1e-3, what="freqemb ")


MOCEIR_TINY = dict(dim=16, levels=4, heads=[1, 2, 4, 8], num_blocks=[1, 1, 1, 2], num_dec_blocks=[1, 1, 1],
                   num_refinement_blocks=1, rank=2, num_experts=4, depth_type="constant", stage_depth=[1, 1, 1],
                   rank_type="spread", topk=1, with_complexity=True, complexity_scale="max")


def test_moceir_whole_network_train_and_eval():
    """BASELINE configs[3] (MoCE-IR) as a whole network at a reduced width: output, auxiliary loss, input gradient and the
    gradient norm of EVERY parameter against the reference (train); batch-1 eval output."""
    import image_restoration_amd.moce_ir as mo
    gold = load("moceir_tiny_train")
    keys = load("moceir_keys")
    net = mo.MoCEIR(**MOCEIR_TINY)
    assert list(net.state_dict()) == [str(k) for k in keys["tiny"]]
    net = _filled(net, 120).to(DEV).train()
    x = seeded_input((2, 3, 64, 64), 1200).to(DEV).requires_grad_(True)
    with injected_noise(1201):
        y = net(x)
        loss = (y - seeded_input((2, 3, 64, 64), 1202).to(DEV)).abs().mean() + 0.01 * net.total_loss
        loss.backward()
    check("y", y, gold, 1e-3); check("dx", x.grad, gold, 2e-3)
    assert abs(float(net.total_loss) - float(gold["total_loss"])) < 1e-4
    assert abs(float(loss) - float(gold["loss"])) < 1e-4
    ref_norms = dict(zip([str(n) for n in gold["grad_names"]], gold["grad_norms"]))
    got = {n: p.grad for n, p in net.named_parameters()}
    worst = 0.0
    for n, rn in ref_norms.items():
        if rn < 0:                       # the reference left this parameter without a gradient (expert nobody routed to)
            assert got[n] is None or float(got[n].norm()) == 0.0, n
            continue
        gn = float(got[n].norm()) if got[n] is not None else 0.0
        worst = max(worst, abs(gn - rn) / max(rn, 1e-6))
        assert abs(gn - rn) <= 2e-3 * max(rn, 1e-6) + 1e-7, (n, gn, rn)
    net.eval()
    with torch.no_grad(), injected_noise(1203):
        ye = net(x.detach()[:1])
    check("y", ye, load("moceir_tiny_eval"), 1e-3)


def test_moceir_base_state_dict_keys_and_size():
    import image_restoration_amd.moce_ir as mo
    keys = load("moceir_keys")
    base = mo.MoCEIR(dim=48, num_blocks=[4, 6, 6, 8], num_dec_blocks=[2, 4, 4], levels=4, heads=[1, 2, 4, 8],
                     num_refinement_blocks=4, topk=1, num_experts=4, rank=2, with_complexity=True, depth_type="constant",
                     stage_depth=[1, 1, 1], rank_type="spread", complexity_scale="max")
    assert list(base.state_dict()) == [str(k) for k in keys["base"]]
    assert sum(p.numel() for p in base.parameters()) == int(keys["base_params"][0])


@pytest.mark.parametrize("p,shape", [(4, (2, 5, 12, 20)), (8, (1, 7, 16, 24)), (16, (2, 3, 40, 33)), (32, (1, 2, 64, 300))])
def test_patch_circconv_vs_fft(p, shape):
    """mi_patch_circconv against irfft2(rfft2 * rfft2) in fp64 (ragged planes, >256-pixel rows), forward and the flipped
    form the gradients use; also with the second operand a channel slice (no copy)."""
    from image_restoration_amd import ops
    import torch.nn.functional as Fn
    b, c, h, w = shape
    x = seeded_input(shape, 5000 + p).to(DEV)
    wide = seeded_input((b, 2 * c, h, w), 5001 + p).to(DEV)
    y = wide[:, c:]

    def ref(a, bb):
        def pat(t):
            t = Fn.pad(t.double().cpu(), (0, (p - w % p) % p, 0, (p - h % p) % p))
            return t.reshape(b, c, t.shape[-2] // p, p, t.shape[-1] // p, p).permute(0, 1, 2, 4, 3, 5)
        o = torch.fft.irfft2(torch.fft.rfft2(pat(a)) * torch.fft.rfft2(pat(bb)), s=(p, p))
        return o.permute(0, 1, 2, 4, 3, 5).reshape(b, c, o.shape[2] * p, o.shape[3] * p)[:, :, :h, :w]
    assert rel(ops.patch_circconv(x, y, p), ref(x, y)) < 2e-5
    # gradient form: d/dx sum(cot * circconv(x, y)) = circconv(cot, y, flip)
    xr = x.detach().double().cpu().requires_grad_(True)
    cot = seeded_input(shape, 5002 + p)
    (ref(xr, y) * cot.double()).sum().backward()
    assert rel(ops.patch_circconv(cot.to(DEV), y, p, flip=True), xr.grad) < 2e-5


AUX_W = 1.0e4      # weight of the auxiliary loss in the objective of the real-plane test: router gradients comparable to the others


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("dim,heads,hw", [(48, 1, (256, 256)), (96, 2, (128, 128)), (192, 4, (64, 64))])
def test_decoder_block_at_real_planes_vs_oracle(dtype, tol, dim, heads, hw):
    """The MoCE DecoderBlock at the plane sizes of BASELINE configs[3] (dim 48 at 256^2, 96 at 128^2, 192 at 64^2): the 7x7
    depthwise kv conv of CrossAttention / FFTAttention, the cross-MDTA row kernels, the patch circular convolution with 32 x 32
    patches and the expert GEMMs at their real tile counts - forward, input gradients and every parameter gradient against the
    oracle (fp32 on the host).  Round-1's fixtures reached these kernels on 16 x 16 planes only."""
    import image_restoration_amd.moce_ir as mo
    H, W = hw
    kw = dict(dim=dim, num_heads=heads, ffn_expansion_factor=2, bias=False, LayerNorm_type="WithBias", expert_layer=mo.FFTAttention,
              complexity_scale="max", rank=2, num_experts=4, top_k=1, depth_type="constant", rank_type="spread", stage_depth=1,
              freq_dim=64, with_complexity=True)
    m = mo.DecoderBlock(**kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("complexity")}
    sd = R.make_state(shapes, 300 + dim)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).train()
    B = 2
    x = seeded_input((B, dim, H, W), 3000 + dim)
    fe = seeded_input((B, 64), 3001 + dim)
    cot = seeded_input((B, dim, H, W), 3002 + dim)
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    with injected_noise(3003 + dim):
        out, aux = m(xg, fe.to(DEV))
    # The objective carries the router's auxiliary loss, as the training step does (train.py:62-71).  Through the MAIN path alone
    # the router weights get NO gradient at top-1: the adapter output of a sample is its expert's output times one gate value, and
    # it enters CrossAttention as the query, which is L2-normalised per channel row - the gate cancels exactly
    # (tests/test_oracle_golden_moce.py::test_router_main_path_gradient_is_exactly_zero_at_top1 shows 1e-17 in fp64).  Round 2
    # compared that round-off-sized "gradient" relatively, saw 120 % and widened a skip threshold; with the aux term the router
    # gradients are real numbers and every parameter is held to the bound, none skipped.
    (out.float() * cot.to(DEV)).sum().add(aux.float() * AUX_W).backward()
    # oracle on the same (dtype-rounded) input, fp32 on the host
    cfg = dict(dim=dim, rank=2, num_experts=4, top_k=1, rank_type="spread", with_complexity=True,
               complexity=m.adapter.routing.complexity.cpu().float())
    xr = xg.detach().float().cpu().requires_grad_(True)
    ps = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    ref, aux_r = MR.decoder_block(xr, fe, ps, heads, cfg, seeded_input((B, 4), 3003 + dim), True)
    ((ref * cot).sum() + aux_r * AUX_W).backward()
    assert rel(out, ref) < tol, ("y", rel(out, ref))
    assert rel(xg.grad, xr.grad) < 3 * tol, ("dx", rel(xg.grad, xr.grad))
    assert abs(float(aux) - float(aux_r)) < 1e-3
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        g_ref = ps[name].grad
        assert g_ref is not None or "experts." in name, name                  # only an un-routed expert has no gradient
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        e = rel(p.grad, g_ref)
        if e > worst[1]:
            worst = (name, e)
    # temperature / router gradients are differences of large sums: 10x the activation bound (as in the Restormer block tests)
    assert worst[1] < 10 * tol, worst


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_grouped_expert_gemm_reads_segments_from_device_tables(dtype, tol):
    """csrc/grouped.hip: the projections of all experts in ONE launch; counts / offsets live on the device (the router's tables),
    ranks differ per expert, one expert gets no rows, ragged plane (N not a multiple of the 64-pixel tile), local and stitched row
    indexing, bias, residual and the transposed (input-gradient) weight use - against per-segment matmuls in fp64."""
    from image_restoration_amd import ops
    R_, Cc, H, W = 7, 48, 9, 21
    N = H * W
    counts = torch.tensor([3, 0, 1, 3], dtype=torch.int32)
    offsets = torch.tensor([0, 3, 3, 4, 7], dtype=torch.int32)
    ranks = [6, 12, 24, 48]
    x = seeded_input((R_, Cc, H, W), 801).to(dtype)
    res = seeded_input((R_, Cc, H, W), 802).to(dtype)
    w0 = [seeded_input((r, Cc), 810 + e) * 0.2 for e, r in enumerate(ranks)]
    w2 = [seeded_input((Cc, r), 820 + e) * 0.2 for e, r in enumerate(ranks)]
    bias = [seeded_input((r,), 830 + e) for e, r in enumerate(ranks)]
    xg, rg = x.to(DEV), res.to(DEV)
    cg, og = counts.to(DEV), offsets.to(DEV)
    w0g, w2g, bg = [w.to(DEV) for w in w0], [w.to(DEV) for w in w2], [b.to(DEV) for b in bias]
    # stage 1: a_e = W0_e x[seg e] + b_e (local rows); stage 2: out[seg e] = W2_e a_e + res[seg e] (stitched rows)
    a = [torch.full((max(int(counts[e]), 1), r, H, W), float("nan"), dtype=dtype, device=DEV) for e, r in enumerate(ranks)]
    ops.grouped_pw_gemm([dict(x=xg, w=w0g[e], bias=bg[e], y=a[e], m=ranks[e], k=Cc, expert=e, y_local=True) for e in range(4)],
                        cg, og, R_, N, dtype)
    out = torch.full((R_, Cc, H, W), float("nan"), dtype=dtype, device=DEV)
    ops.grouped_pw_gemm([dict(x=a[e], w=w2g[e], r=rg, y=out, m=Cc, k=ranks[e], expert=e, x_local=True) for e in range(4)],
                        cg, og, R_, N, dtype)
    # input-gradient form: dx[seg e] = W0_e^T a_e
    dx = torch.full((R_, Cc, H, W), float("nan"), dtype=dtype, device=DEV)
    ops.grouped_pw_gemm([dict(x=a[e], w=w0g[e], transposed=True, y=dx, m=Cc, k=ranks[e], expert=e, x_local=True) for e in range(4)],
                        cg, og, R_, N, dtype)
    for e in range(4):
        n, o = int(counts[e]), int(offsets[e])
        if not n:
            assert bool(torch.isnan(a[e].float()).all())                     # an expert without rows writes nothing
            continue
        xs = x[o:o + n].double().reshape(n, Cc, N)
        ar = torch.einsum("mk,nkp->nmp", w0[e].double(), xs) + bias[e].double()[None, :, None]
        assert rel(a[e][:n].reshape(n, ranks[e], N), ar) < tol, (e, "a")
        a_used = a[e][:n].double().cpu().reshape(n, ranks[e], N)               # what stage 2 actually read (dtype-rounded)
        outr = torch.einsum("mk,nkp->nmp", w2[e].double(), a_used) + res[o:o + n].double().reshape(n, Cc, N)
        assert rel(out[o:o + n].reshape(n, Cc, N), outr) < tol, (e, "out")
        dxr = torch.einsum("km,nkp->nmp", w0[e].double(), a_used)
        assert rel(dx[o:o + n].reshape(n, Cc, N), dxr) < tol, (e, "dx")
    assert not bool(torch.isnan(out.float()).any()) and not bool(torch.isnan(dx.float()).any())


@pytest.mark.parametrize("top_k", [1, 2])
def test_adapter_capacity_dispatch_equals_ragged_dispatch(top_k):
    """AdapterLayer without the host read-back (capacity mode: every expert's buffers hold all B * k rows, the grouped launches
    take the true segment sizes from the router's device tables, bodies run over zero rows beyond them) against the ragged mode
    that reads the E sizes back like the reference (moce_ir.py:88): same output, same input gradients, same parameter gradients."""
    from image_restoration_amd import moce_ir
    torch.manual_seed(3)
    dim, E = 48, 4
    ad = moce_ir.AdapterLayer(dim, rank=4, num_experts=E, top_k=top_k, expert_layer=moce_ir.FFTAttention, stage_depth=1,
                              depth_type="constant", rank_type="spread", freq_dim=32, with_complexity=True).to(DEV).train()
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn((6, dim, 32, 32), generator=g).to(DEV)
    sh0 = torch.randn((6, dim, 32, 32), generator=g).to(DEV)
    fe = torch.randn((6, 32), generator=g).to(DEV)
    res = {}
    for mode in ("capacity", "ragged"):
        ad.dispatch = mode
        for p in ad.parameters():
            p.grad = None
        torch.manual_seed(17)                      # the router draws its noise from the global generator
        x, sh = x0.clone().requires_grad_(True), sh0.clone().requires_grad_(True)
        y = ad(x, fe, sh)
        (y.float().square().mean() + 0.1 * ad.loss).backward()
        res[mode] = [y.detach(), x.grad, sh.grad] + [p.grad.clone() if p.grad is not None else None for p in ad.parameters()]
    assert int(ad.routing.tables.counts.sum()) == 6 * top_k
    for a, b in zip(res["capacity"], res["ragged"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel(a, b) < 2e-5, rel(a, b)
