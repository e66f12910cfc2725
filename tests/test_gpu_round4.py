"""Round-4 GPU cases: gradient accumulation with deferred sums (same-output jobs), HIP-graph capture of whole training steps
(regression guard for the hipStreamEndCapture fault of round 3), the trainer's grads_ready() contract."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")


def M():
    import image_restoration_amd as m
    return m


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_micro_batches_under_no_sync_with_deferred_sums(monkeypatch):
    """three micro-batches between zero_grad() and reduce_gradients() record three sums per gradient.  The
    flush must run them one after the other (generations), not as concurrent read-modify-writes of one launch: the accumulated
    gradient equals the one with the deferral off, and two deferred runs are bit-identical."""
    m = M()
    from image_restoration_amd import ops
    from image_restoration_amd.configs import RESTORMER_TINY
    from image_restoration_amd.trainer import FlatTrainer
    g = torch.Generator().manual_seed(9)
    xs = [torch.rand((2, 3, 64, 64), generator=g).to(DEV).to(torch.bfloat16) for _ in range(3)]

    def run(defer_mb):
        monkeypatch.setenv("MI_DEFER_MB", str(defer_mb))
        torch.manual_seed(3)
        net = m.Restormer(**RESTORMER_TINY).to(DEV)
        tr = FlatTrainer(net, lr=1e-2)
        try:
            tr.zero_grad()
            with tr.no_sync():
                for x in xs[:2]:
                    net(x).float().abs().mean().backward()
            net(xs[2]).float().abs().mean().backward()
            if defer_mb > 0:
                assert ops.deferred_pending() > 0
            tr.reduce_gradients()
            assert ops.deferred_pending() == 0
            return tr.flat_g.clone()
        finally:
            tr.close()
    g1, g2, g0 = run(256), run(256), run(0)
    assert torch.equal(g1, g2)
    assert rel(g1, g0) < 1e-5, rel(g1, g0)
    # one micro-batch alone gives a different (smaller) gradient: the comparison above is not vacuous
    assert float(g0.abs().sum()) > 0


def test_optimizer_step_without_reduce_gradients_sees_final_gradients(monkeypatch):
    """a single-GPU loop that skips reduce_gradients() steps on complete gradients (optimizer_step flushes),
    and grads_ready() is the call for any other reader of flat_g."""
    m = M()
    from image_restoration_amd import ops
    from image_restoration_amd.configs import RESTORMER_TINY
    from image_restoration_amd.trainer import FlatTrainer
    x = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(5)).to(DEV).to(torch.bfloat16)

    def run(call_reduce):
        monkeypatch.setenv("MI_DEFER_MB", "64")
        torch.manual_seed(3)
        net = m.Restormer(**RESTORMER_TINY).to(DEV)
        tr = FlatTrainer(net, lr=1e-2)
        try:
            tr.zero_grad()
            net(x).float().abs().mean().backward()
            assert ops.deferred_pending() > 0
            if call_reduce:
                tr.reduce_gradients()
            tr.optimizer_step()
            assert ops.deferred_pending() == 0
            return tr.flat_p.clone(), tr.flat_g.clone()
        finally:
            tr.close()
    p1, g1 = run(True)
    p2, g2 = run(False)
    assert torch.equal(g1, g2) and torch.equal(p1, p2)


CHILD = textwrap.dedent(r'''
    import os, sys, torch
    sys.path.insert(0, os.getcwd())
    import image_restoration_amd as m
    from image_restoration_amd import configs, ops, moce_ir
    from image_restoration_amd.trainer import FlatTrainer
    case = sys.argv[1]
    dev = "cuda"
    torch.manual_seed(0)
    if case == "restormer_tiny":
        net = m.Restormer(**configs.RESTORMER_TINY).to(dev)
        x = torch.rand(2, 3, 64, 64, device=dev).to(torch.bfloat16)
        aux = lambda: 0.0
    else:
        os.environ["MI_MOCE_DISPATCH"] = "capacity"          # segment sizes stay on the device: nothing to read back mid-capture
        m.reload_env()
        net = moce_ir.MoCEIR(**configs.MOCEIR_TINY).to(dev).train()
        x = torch.rand(4, 3, 64, 64, device=dev).to(torch.bfloat16)
        aux = lambda: 0.01 * net.total_loss
    tr = FlatTrainer(net, lr=1e-4)                           # DEFAULT trainer: deferred sums on
    assert tr._defer_token is not None
    losses = []
    def step():
        tr.zero_grad()
        out = net(x)
        loss = out.float().abs().mean() + aux()
        loss.backward()
        tr.reduce_gradients()
        tr.optimizer_step(use_dev_scalars=True)
        losses.append(loss.detach())
    graph = tr.capture_step(step, warmup=2)
    p_before = tr.flat_p.clone()
    for _ in range(3):
        tr.replay_step(graph)
    torch.cuda.synchronize()
    moved = float((tr.flat_p - p_before).abs().max())
    assert moved > 0, "replays did not update the parameters"
    assert torch.isfinite(tr.flat_p).all()
    # an eager step after the replays still works (deferred sums back on) and keeps the loss finite
    tr.set_step_scalars(tr.step_count + 1)
    step()
    torch.cuda.synchronize()
    assert torch.isfinite(losses[-1]).all()
    print("CAPTURE_OK", case, moved)
''')


@pytest.mark.parametrize("case", ["restormer_tiny", "moce_tiny"])
def test_whole_training_step_captures_and_replays_as_a_hip_graph(case, tmp_path):
    """One full step (zero_grad, forward, loss, backward, reduce, AdamW) of Restormer-tiny and of a small MoCE-IR (router
    backward included) captured by FlatTrainer.capture_step with a DEFAULT trainer and replayed, in a fresh child process (a
    failed hipStreamEndCapture takes the process down: round 3's r3ad / r3ae logs)."""
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ)
    env.pop("MI_DEFER_MB", None)
    res = subprocess.run([sys.executable, str(script), case], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert res.returncode == 0 and "CAPTURE_OK" in res.stdout, res.stdout[-3000:]


def test_deferred_flush_refuses_a_capturing_stream():
    """with sums pending, mi_deferred_flush on a capturing stream returns an error instead of baking a host
    copy into the graph; producers called on a capturing stream do not defer."""
    m = M()
    from image_restoration_amd import ops
    from image_restoration_amd.trainer import FlatTrainer
    blk = m.TransformerBlock(48, 1, 2.66, False, "WithBias").to(DEV)
    tr = FlatTrainer(blk, lr=1e-2)
    try:
        x = torch.randn(2, 48, 32, 32, device=DEV).to(torch.bfloat16).requires_grad_(True)
        tr.zero_grad()
        blk(x).float().square().mean().backward()
        assert ops.deferred_pending() > 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            g.capture_begin()
            try:
                with pytest.raises(RuntimeError, match="captured"):
                    ops.deferred_flush()
            finally:
                g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        assert ops.deferred_pending() > 0
        tr.reduce_gradients()
        assert ops.deferred_pending() == 0
    finally:
        tr.close()


# ----------------------------------------------------------------------------------------------- fused half-blocks, real planes
def _seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("c,heads,hw,batch", [(48, 1, 256, 8), (96, 2, 128, 32), (96, 1, 256, 8)])
def test_fused_mdta_at_the_training_plane_vs_oracle_and_chain(c, heads, hw, batch):
    """Round-3 verdict, weak 1(c): the fused pass A (C = 48: the fourth form - depthwise conv on the matrix cores, wave-local Gram -
    C = 96: the round-3 form) in the regime it ships in: 8 x C x 256^2 / 128^2, one persistent workgroup per CU walking many
    tiles.  Against the fp64 oracle on the first two images (host time) and against the unfused chain on all of them."""
    m = M()
    from image_restoration_amd import ops
    from oracle import restormer_ref as R
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=400 + c + heads)
    x = _seeded((batch, c, hw, hw), 4100 + c).to(DEV).to(torch.bfloat16)
    ln = (sd["norm1.body.weight"].to(DEV), sd["norm1.body.bias"].to(DEV))
    keys = ["attn.temperature", "attn.qkv.weight", "attn.qkv.bias", "attn.qkv_dwconv.weight", "attn.qkv_dwconv.bias",
            "attn.project_out.weight", "attn.project_out.bias"]
    att = tuple(sd[k].to(DEV).float().contiguous() if k in sd else None for k in keys)
    assert ops.mdta_fused_ok(x, heads, 3) and ops.mdta_fused_pays(x, heads, 3)
    pack = ops.mdta_fused_pack(x, heads, ln[0], ln[1], att)
    y, mean, rstd = ops.mdta_fused_fwd(x, pack, att, heads, True, x, want_stats=True)
    xn, mean_r, rstd_r = ops.ln_fwd(x, ln[0], ln[1], True, want_stats=True)
    chain, _ = ops.mdta_fwd(xn, x, att, heads, False)
    assert rel(mean, mean_r) < 1e-5 and rel(rstd, rstd_r) < 1e-4
    x2 = x[:2].float().cpu().double()
    d = {k: v.double() for k, v in sd.items()}
    xn2 = R.layernorm_nchw(x2, d["norm1.body.weight"], d.get("norm1.body.bias"), "WithBias")
    ref = x2 + R.mdta(xn2, d["attn.temperature"], d["attn.qkv.weight"], d["attn.qkv_dwconv.weight"], d["attn.project_out.weight"], heads,
                      d.get("attn.qkv.bias"), d.get("attn.qkv_dwconv.bias"), d.get("attn.project_out.bias"))
    e_or, e_ch = rel(y[:2], ref), rel(chain[:2], ref)
    assert e_or < 2e-2, (e_or, e_ch)
    assert e_or < 1.5 * e_ch + 4e-3, (e_or, e_ch)
    assert rel(y, chain.double()) < 2e-2                      # every image, against the chain


@pytest.mark.parametrize("c,hidden,hw,batch", [(48, 127, 256, 8), (96, 255, 128, 8)])
def test_fused_gdfn_at_the_training_plane_vs_oracle_and_chain(c, hidden, hw, batch):
    """The one-launch LN + GDFN half-block (C = 48: the fourth form) in its persistent regime, inference and SAVE forms: output against
    the fp64 oracle (two images) and the chain (all), and the saved blob against what the chain saves."""
    m = M()
    from image_restoration_amd import ops
    from oracle import restormer_ref as R
    sd = R.make_block_state(c, 1, 2.66, False, "WithBias", seed=470 + c)
    y = _seeded((batch, c, hw, hw), 4700 + c).to(DEV).to(torch.bfloat16)
    ln_w, ln_b = sd["norm2.body.weight"].to(DEV), sd["norm2.body.bias"].to(DEV)
    keys = ["ffn.project_in.weight", "ffn.project_in.bias", "ffn.dwconv.weight", "ffn.dwconv.bias", "ffn.project_out.weight",
            "ffn.project_out.bias"]
    params = tuple(sd[k].to(DEV).float().contiguous() if k in sd else None for k in keys)
    assert params[0].shape[0] == 2 * hidden
    pack = ops.gdfn_fused_pack(y, ln_w, ln_b, params)
    out, mean, rstd = ops.gdfn_fused_fwd(y, pack, hidden, True, want_stats=True)
    chain = ops.gdfn_fwd(y, y, params, True, ln=(ln_w, ln_b, True))
    chain_out = chain[0] if isinstance(chain, (tuple, list)) else chain
    y2 = y[:2].float().cpu().double()
    d = {k: v.double() for k, v in sd.items()}
    yn2 = R.layernorm_nchw(y2, d["norm2.body.weight"], d.get("norm2.body.bias"), "WithBias")
    ref = y2 + R.gdfn(yn2, d["ffn.project_in.weight"], d["ffn.dwconv.weight"], d["ffn.project_out.weight"], d.get("ffn.project_in.bias"),
                      d.get("ffn.dwconv.bias"), d.get("ffn.project_out.bias"))
    e_or, e_ch = rel(out[:2], ref), rel(chain_out[:2], ref)
    assert e_or < 2e-2, (e_or, e_ch)
    assert e_or < 1.5 * e_ch + 4e-3, (e_or, e_ch)
    assert rel(out, chain_out.double()) < 2e-2
    out_t, saved, mean_t, rstd_t = ops.gdfn_fused_fwd_train(y, pack, hidden, True)
    assert rel(out_t, out.double()) < 1e-2 and rel(mean_t, mean) < 1e-5 and rel(rstd_t, rstd) < 1e-4


def test_forward_gelu_of_the_bf16_kernels_stays_within_its_stated_bound():
    """common.h gelu_fwd<bf16>: x sigmoid(1.6 x (1 + 0.0435 x^2)), |difference to the erf form| <= 3e-4 (stated in the header);
    checked here on the formula itself in fp64 over a dense grid (the kernels' outputs are covered by the oracle tests above)."""
    import math
    x = torch.linspace(-12, 12, 480001, dtype=torch.float64)
    exact = x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))
    u = x * (-2.3083120 - 0.1004116 * x * x)
    fast = x / (1 + torch.exp2(u))
    assert float((fast - exact).abs().max()) < 3.0e-4


@pytest.mark.parametrize("c,heads,shape", [(40, 1, (2, 40, 8, 16)), (72, 3, (2, 72, 8, 16)), (30, 3, (1, 30, 16, 8)),
                                           (112, 1, (1, 112, 8, 16)), (120, 1, (2, 120, 8, 8)), (128, 2, (1, 128, 8, 8)),
                                           (384, 8, (3, 384, 4, 8))])
def test_attention_small_kernels_at_padded_and_wide_heads(c, heads, shape):
    """The c x c side of MDTA (attn_fold: DPP-row softmax + fp32-MFMA fold; the one-launch attention backward) pads channels per
    head to a multiple of 16 in LDS and masks at the stores: channels per head that are NOT multiples of 16 (40, 24, 10), the
    widest tiles (112, 120 -> 8 x 8 fragments), C not a multiple of 4 (scalar path of the transposed stores) and several W_o row
    chunks per workgroup (384 / 8 heads at 3 images) against the fp64 oracle (Restormer.py:111-131), forward, dx and every
    parameter gradient, bound 5e-5."""
    import image_restoration_amd as m
    from oracle import restormer_ref as R
    from oracle.fixtures import seeded_input
    sd = R.make_block_state(c, heads, 2.66, True, "WithBias", seed=3 * c + heads)
    blk = m.TransformerBlock(c, heads, 2.66, True, "WithBias").to("cuda")
    blk.load_state_dict(sd)
    x, cot = seeded_input(shape, 4100 + c), seeded_input(shape, 4101 + c)
    xg = x.to("cuda").requires_grad_(True)
    y = blk(xg)
    y.backward(cot.to("cuda"))
    xr = x.double().requires_grad_(True)
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.transformer_block(xr, sdr, heads, "WithBias")
    yr.backward(cot.double())
    assert rel(y, yr) < 5e-5
    assert rel(xg.grad, xr.grad) < 5e-5
    for k, p in blk.named_parameters():
        assert rel(p.grad, sdr[k].grad) < 5e-5, k
