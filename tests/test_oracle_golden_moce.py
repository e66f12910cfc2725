"""Pins oracle/moce_ref.py against golden vectors captured from the imported reference moce_ir.py /
AdaIR-main/net/model.py (tools/capture_golden_moce.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import moce_ref as MR
from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input

F64 = torch.float64

CROSS_KEYS = ["temperature", "q.weight", "q.bias", "q_dwconv.weight", "q_dwconv.bias", "kv.weight", "kv.bias",
              "kv_dwconv.weight", "kv_dwconv.bias", "project_out.weight", "project_out.bias"]


def cross_shapes(c, heads, bias, ks_kv):
    s = {"temperature": (heads, 1, 1), "q.weight": (c, c, 1, 1), "q_dwconv.weight": (c, 1, 3, 3),
         "kv.weight": (2 * c, c, 1, 1), "kv_dwconv.weight": (2 * c, 1, ks_kv, ks_kv), "project_out.weight": (c, c, 1, 1)}
    if bias:
        s.update({"q.bias": (c,), "q_dwconv.bias": (c,), "kv.bias": (2 * c,), "kv_dwconv.bias": (2 * c,),
                  "project_out.bias": (c,)})
    return {k: s[k] for k in CROSS_KEYS if k in s}   # reference registration order


def _run(fn, inputs, sd, seed):
    ins = [t.clone().requires_grad_(True) for t in inputs]
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = fn(*ins, ps)
    y0 = y[0] if isinstance(y, tuple) else y
    y0.backward(seeded_input(tuple(y0.shape), seed + 1000, y0.dtype))
    return y, [t.grad for t in ins], {k: v.grad for k, v in ps.items() if v.grad is not None}


@pytest.mark.parametrize("tag,c,heads,seed,xs", [("moce_cross_c48h1", 48, 1, 60 + 48, 600 + 48),
                                                  ("moce_cross_c96h2", 96, 2, 60 + 96, 600 + 96)])
def test_cross_attention_moce(tag, c, heads, seed, xs):
    sd = R.make_state(cross_shapes(c, heads, True, 7), seed, F64)
    x, y = seeded_input((2, c, 16, 16), xs, F64), seeded_input((2, c, 16, 16), xs + 1, F64)
    out, (dx, dy), g = _run(lambda a, b, p: MR.cross_attention(a, b, p, heads), [x, y], sd, 610)
    gold = load(tag)
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6); check("dy", dy, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


def test_cross_attention_adair():
    sd = R.make_state(cross_shapes(48, 4, False, 3), 65, F64)
    x, y = seeded_input((2, 48, 16, 16), 650, F64), seeded_input((2, 48, 16, 16), 651, F64)
    out, (dx, dy), g = _run(lambda a, b, p: MR.cross_attention(a, b, p, 4), [x, y], sd, 660)
    gold = load("adair_cross_c48h4")
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6); check("dy", dy, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


@pytest.mark.parametrize("k", [1, 2])
def test_routing_and_dispatch(k):
    gold = load(f"moce_routing_k{k}")
    sd = R.make_state({"gate.2.weight": (4, 48), "freq_gate.weight": (4, 64)}, 70 + k, F64)
    comp = torch.tensor([18840., 42288., 103008., 279744.], dtype=F64)
    comp = comp / comp.max()
    x, fe = seeded_input((8, 48, 8, 8), 700, F64), seeded_input((8, 64), 701, F64)
    noise = seeded_input((8, 4), 702, F64)
    gates, idx, vals, aux = MR.routing(x, fe, sd, k, noise, True, comp, True)
    check("gates", gates, gold, 2e-6)
    assert np.array_equal(idx.numpy(), gold["idx"])
    assert abs(float(aux) - float(gold["aux"])) < 1e-7  # complexity buffer was normalised in fp32
    rows = MR.dispatch_indices(gates)
    assert [len(r) for r in rows] == list(gold["part_sizes"])
    comb = torch.zeros(8, 48, 8, 8)
    for e, r in enumerate(rows):
        if r:
            rr = torch.tensor(r)
            comb = comb.index_add(0, rr, (x[rr] * (e + 1) * gates[rr, e].view(-1, 1, 1, 1)).float())
    check("combined", comb, gold, 2e-6)


def encoder_shapes(c, heads, ffn, bias):
    b = R._block_shapes(c, heads, ffn, bias, "WithBias")
    ren = {"norm1": "norms.0", "norm2": "norms.1", "attn": "mixer", "ffn": "ffn"}
    out = {}
    for pre in ("norm1", "norm2", "attn", "ffn"):          # reference registration order: norms, mixer, ffn
        for k, v in b.items():
            if k.startswith(pre + "."):
                out[ren[pre] + k[len(pre):]] = v
    return out


def test_encoder_block():
    sd = R.make_state(encoder_shapes(48, 2, 2, True), 75, F64)
    x = seeded_input((2, 48, 16, 16), 750, F64)
    out, (dx,), g = _run(lambda a, p: MR.encoder_block(a, p, 2), [x], sd, 760)
    gold = load("moce_encoder_c48h2")
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


def decoder_state(dtype):
    """Same seeded fill as the capture: shapes in the reference module's registration order (taken from the product
    module, whose state_dict layout is asserted equal to the reference's in test_cabi)."""
    import image_restoration_amd.moce_ir as mo
    kw = dict(dim=48, num_heads=1, ffn_expansion_factor=2, bias=False, LayerNorm_type="WithBias", expert_layer=mo.FFTAttention,
              complexity_scale="max", rank=2, num_experts=4, top_k=1, depth_type="constant", rank_type="spread", stage_depth=1,
              freq_dim=64, with_complexity=True)
    m = mo.DecoderBlock(**kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("complexity")}
    return m, R.make_state(shapes, 80, dtype)


def test_decoder_block_train_and_eval():
    gold = load("moce_decoder_train")
    _, sd = decoder_state(F64)
    cfg = dict(dim=48, rank=2, num_experts=4, top_k=1, rank_type="spread", with_complexity=True,
               complexity=torch.tensor(gold["complexity"], dtype=F64))
    x, fe = seeded_input((4, 48, 16, 16), 800, F64), seeded_input((4, 64), 801, F64)
    noise = seeded_input((4, 4), 802, F64)
    (out, aux), (dx, dfe), g = _run(lambda a, b, p: MR.decoder_block(a, b, p, 1, cfg, noise, True), [x, fe], sd, 810)
    # the reference ran in fp32 (its combine is float32): 2e-4 covers fp32 round-off of the whole block
    check("y", out, gold, 2e-4); check("dx", dx, gold, 5e-4)
    assert abs(float(aux) - float(gold["aux"])) < 1e-5
    for k, v in g.items():
        if float(gold["g_" + k + ".l2"]) < 1e-6:
            # with top_k = 1 the adapter output of a sample is scaled by ONE gate value and then L2-normalised as the
            # cross-attention query, so the router weights get no gradient through the main path (fp32 noise in the fixture)
            assert float(v.norm()) < 1e-6, k
            continue
        check("g_" + k, v, gold, 2e-3, what="decoder ")
    gold_e = load("moce_decoder_eval")
    with torch.no_grad():
        out_e, aux_e = MR.decoder_block(x[:1], fe[:1], sd, 1, cfg, seeded_input((1, 4), 803, F64), False)
    check("y", out_e, gold_e, 2e-4)
    assert aux_e == 0


# ------------------------------------------------------------------------------------------------ round-2 fixtures
def _module_shapes(mod):
    return {k: tuple(v.shape) for k, v in mod.state_dict().items() if not k.endswith("complexity")}


@pytest.mark.parametrize("tag,r,p,shape", [("r12p8", 12, 8, (2, 12, 16, 16)), ("r24p16", 24, 16, (1, 24, 20, 12))])
def test_fft_attention_standalone(tag, r, p, shape):
    import image_restoration_amd.moce_ir as mo
    sd = R.make_state(_module_shapes(mo.FFTAttention(r, patch_size=p, kernel_size=3)), 90 + r, F64)
    out, (dx,), g = _run(lambda a, ps: MR.fft_attention(a, ps, p), [seeded_input(shape, 900 + r, F64)], sd, 910 + r)
    gold = load(f"moce_fftattn_{tag}")
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


def test_mod_expert_standalone():
    import image_restoration_amd.moce_ir as mo
    sd = R.make_state(_module_shapes(mo.ModExpert(48, rank=12, func=mo.FFTAttention, depth=1, patch_size=8, kernel_size=5)),
                      95, F64)
    x, sh = seeded_input((2, 48, 16, 16), 950, F64), seeded_input((2, 48, 16, 16), 951, F64)
    out, (dx, dsh), g = _run(lambda a, b, ps: MR.mod_expert(a, b, ps, 8), [x, sh], sd, 960)
    gold = load("moce_modexpert_c48r12")
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6); check("dshared", dsh, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


@pytest.mark.parametrize("k", [1, 2])
def test_adapter_layer_all_experts(k):
    import image_restoration_amd.moce_ir as mo
    gold = load(f"moce_adapter_k{k}")
    al = mo.AdapterLayer(48, rank=2, num_experts=4, top_k=k, expert_layer=mo.FFTAttention, stage_depth=1,
                         depth_type="constant", rank_type="spread", freq_dim=64, with_complexity=True, complexity_scale="max")
    sd = R.make_state(_module_shapes(al), 100 + k, F64)
    cfg = dict(dim=48, rank=2, num_experts=4, top_k=k, rank_type="spread", with_complexity=True,
               complexity=al.routing.complexity.double())
    ins = [seeded_input((8, 48, 16, 16), 1000, F64), seeded_input((8, 64), 1001, F64), seeded_input((8, 48, 16, 16), 1002, F64)]
    noise = seeded_input((8, 4), int(gold["noise_seed"]), F64)
    (out, aux), (dx, dfe, dsh), g = _run(lambda a, b, c, ps: MR.adapter_layer(a, b, c, ps, cfg, noise, True), ins, sd, 1010 + k)
    # the reference ran in fp32 (its combine buffer is float32)
    check("y", out, gold, 2e-4); check("dx", dx, gold, 5e-4); check("dshared", dsh, gold, 5e-4)
    assert abs(float(aux) - float(gold["aux"])) < 1e-5


@pytest.mark.parametrize("k", [1, 2])
def test_router_gradients(k):
    gold = load(f"moce_router_grads_k{k}")
    sd = R.make_state({"gate.2.weight": (4, 48), "freq_gate.weight": (4, 64)}, 70 + k, F64)
    comp = torch.tensor([18840., 42288., 103008., 279744.], dtype=F64)
    comp = comp / comp.max()
    x = seeded_input((8, 48, 8, 8), 700, F64).requires_grad_(True)
    fe = seeded_input((8, 64), 701, F64).requires_grad_(True)
    ps = {kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    gates, idx, vals, aux = MR.routing(x, fe, ps, k, seeded_input((8, 4), 702, F64), True, comp, True)
    ((gates * seeded_input((8, 4), 703, F64)).sum() + aux).backward()
    check("gates", gates, gold, 2e-6)
    check("dx", x.grad, gold, 2e-6); check("dfe", fe.grad, gold, 2e-6)
    check("g_gate", ps["gate.2.weight"].grad, gold, 2e-6); check("g_freq", ps["freq_gate.weight"].grad, gold, 2e-6)


def test_frequency_embedding():
    import image_restoration_amd.moce_ir as mo
    sd = R.make_state(_module_shapes(mo.FrequencyEmbedding(64)), 110, F64)
    out, (dx,), g = _run(lambda a, ps: MR.frequency_embedding(a, ps), [seeded_input((2, 64, 8, 8), 1100, F64)], sd, 1110)
    gold = load("moce_freqemb_d64")
    check("y", out, gold, 2e-6); check("dx", dx, gold, 2e-6)
    for k, v in g.items():
        check("g_" + k, v, gold, 2e-6)


MOCEIR_TINY = dict(dim=16, levels=4, heads=[1, 2, 4, 8], num_blocks=[1, 1, 1, 2], num_dec_blocks=[1, 1, 1],
                   num_refinement_blocks=1, rank=2, num_experts=4, depth_type="constant", stage_depth=[1, 1, 1],
                   rank_type="spread", topk=1, with_complexity=True, complexity_scale="max")


def test_moceir_whole_network_oracle():
    """oracle.moce_ref.moceir_forward (the whole MoCE-IR network: glue, encoder / latent groups, frequency embedding, MoCE decoder,
    refinement, aux-loss bookkeeping) against the goldens captured from the reference's MoCEIR at a reduced width: output,
    total_loss, loss, input gradient, every parameter-gradient norm (train) and the batch-1 eval output.  This is what pins the
    oracle the GPU suite holds the BASE configuration (BASELINE configs[3]) against."""
    import image_restoration_amd.moce_ir as mo
    gold = load("moceir_tiny_train")
    net = mo.MoCEIR(**MOCEIR_TINY)
    sd = R.make_state(_module_shapes(net), 120, torch.float32)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = seeded_input((2, 3, 64, 64), 1200).requires_grad_(True)
    noise = seeded_input((2, 4), 1201, F64).float()
    y, total = MR.moceir_forward(x, ps, MOCEIR_TINY, noise, True)
    loss = (y - seeded_input((2, 3, 64, 64), 1202)).abs().mean() + 0.01 * total
    loss.backward()
    check("y", y, gold, 2e-4); check("dx", x.grad, gold, 1e-3)
    assert abs(float(total) - float(gold["total_loss"])) < 1e-5
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    ref_norms = dict(zip([str(n) for n in gold["grad_names"]], gold["grad_norms"]))
    for n, rn in ref_norms.items():
        g = ps[n].grad
        if rn < 0:
            assert g is None or float(g.norm()) == 0.0, n
            continue
        gn = float(g.norm()) if g is not None else 0.0
        assert abs(gn - rn) <= 1e-3 * max(rn, 1e-6) + 1e-7, (n, gn, rn)
    with torch.no_grad():
        ye, te = MR.moceir_forward(x.detach()[:1], sd, MOCEIR_TINY, seeded_input((1, 4), 1203, F64).float(), False)
    check("y", ye, load("moceir_tiny_eval"), 2e-4)
    assert te == 0


def test_router_main_path_gradient_is_exactly_zero_at_top1():
    """Why tests/test_gpu_moce.py's real-plane DecoderBlock test puts the auxiliary loss into its objective: at top-1 (bias-free
    block, the base configuration) a sample's adapter output is ONE gate value times its expert's output, proj_out and the
    CrossAttention query path (1x1 conv, depthwise conv) are linear, and the query is then L2-normalised per channel row - the
    gate cancels.  In fp64 the router weights' gradient through the main path is zero to round-off (relative to the other
    gradients), so any fp32 / bf16 evaluation of it is pure cancellation noise and cannot be compared relatively."""
    import image_restoration_amd.moce_ir as mo
    dim, heads, B = 48, 1, 4
    m = mo.DecoderBlock(dim=dim, num_heads=heads, ffn_expansion_factor=2, bias=False, LayerNorm_type="WithBias",
                        expert_layer=mo.FFTAttention, complexity_scale="max", rank=2, num_experts=4, top_k=1, depth_type="constant",
                        rank_type="spread", stage_depth=1, freq_dim=64, with_complexity=True)
    sd = R.make_state(_module_shapes(m), 348, F64)
    cfg = dict(dim=dim, rank=2, num_experts=4, top_k=1, rank_type="spread", with_complexity=True,
               complexity=m.adapter.routing.complexity.double())
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = seeded_input((B, dim, 16, 16), 3480, F64)
    out, aux = MR.decoder_block(x, seeded_input((B, 64), 3481, F64), ps, heads, cfg, seeded_input((B, 4), 3483, F64), True)
    (out * seeded_input((B, dim, 16, 16), 3482, F64)).sum().backward()
    biggest = max(float(v.grad.abs().max()) for v in ps.values() if v.grad is not None)
    for k in ("adapter.routing.gate.2.weight", "adapter.routing.freq_gate.weight"):
        g = ps[k].grad
        assert g is None or float(g.abs().max()) < 1e-12 * biggest, (k, float(g.abs().max()), biggest)
