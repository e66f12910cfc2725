"""Drop-in replacements for the block classes of the reference's ``Restormer.py``.

Same class names, constructor arguments, parameter/buffer names and shapes (``state_dict``
interchangeable) and ``forward`` signatures as Restormer.py:25-284; ``forward`` runs
the gfx950 kernels through the C-ABI instead of ATen ops.  The conv/LayerNorm submodules are kept
only as parameter containers (so initialisation and key names are the reference's); they are never
called.  Activations may be float32 (exact-fp32 MFMA path, the parity path) or bfloat16 (bf16 MFMA,
fp32 accumulate); parameters and their gradients stay float32.

Gradient accumulation: if every parameter of a module carries a ``main_grad`` attribute (a float32
tensor of the parameter's shape, e.g. a view into a flat DDP bucket), backward accumulates into it
in place and reports no autograd gradient for the parameters; otherwise ordinary ``.grad`` flow.
"""
from __future__ import annotations

import math
import numbers
import os
import threading
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops

Tensor = torch.Tensor


def _main_grads(params: Sequence[Optional[Tensor]]) -> Optional[List[Optional[Tensor]]]:
    """main_grad buffers if every present parameter has one, else None."""
    out = []
    for p in params:
        if p is None:
            out.append(None)
            continue
        mg = getattr(p, "main_grad", None)
        if mg is None:
            return None
        out.append(mg)
    return out


def _fresh_grads(params: Sequence[Optional[Tensor]]) -> List[Optional[Tensor]]:
    return [None if p is None else torch.empty_like(p) for p in params]


# ======================================================================================
# autograd glue
# ======================================================================================
# Inside Function.forward grad mode is always off and ctx.needs_input_grad ignores torch.no_grad(), so the caller's grad
# mode is recorded right before .apply(): under no_grad nothing is saved for backward (no blobs, no LN statistics).
_tls = threading.local()


def _apply(fn, *args):
    _tls.grad = torch.is_grad_enabled()
    return fn.apply(*args)


def _grad_mode() -> bool:
    return getattr(_tls, "grad", True)


def _use_torch_ops(x: Tensor) -> bool:
    """Which door the modules take to the kernels.  The torch.library custom ops (torch_ops.py: ``mi_restore::*_fwd`` / ``_bwd``
    with fake implementations) are what a tracer needs, so they are taken while torch.compile is tracing and whenever
    MI_TORCH_OPS=1; eager execution takes the bare autograd.Function nodes - the same ``_block_forward`` / ``_block_backward``
    underneath, without the dispatcher's per-call cost (MoCE-IR base, 3 500 launches per step, host-bound: 64.7 -> 51.8 ms per
    step; Restormer bs 32, GPU-bound: no difference).  MI_TORCH_OPS=0 forces the direct route."""
    if not x.is_cuda:
        return False
    e = ops.env("MI_TORCH_OPS")
    if e is not None:
        return e != "0"
    return bool(torch.compiler.is_compiling())


def _torch_ops():
    from . import torch_ops
    return torch_ops


def block_apply(x: Tensor, heads: int, params) -> Tensor:
    """One TransformerBlock / EncoderBlock with autograd: through mi_restore::transformer_block (default) or _BlockFn."""
    if _use_torch_ops(x):
        return _torch_ops().transformer_block(x, heads, params)
    return _apply(_BlockFn, x, heads, *params)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        need = _grad_mode() and any(ctx.needs_input_grad)
        y, mean, rstd = ops.ln_fwd(x, weight, bias, bias is not None, want_stats=need)
        if need:
            ctx.save_for_backward(x, weight, mean, rstd)
            ctx.with_bias = bias is not None
            ctx.mg = _main_grads((weight, bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        acc = ctx.mg is not None
        dw, db = ctx.mg if acc else (torch.empty_like(weight), torch.empty_like(weight) if ctx.with_bias else None)
        dx = ops.ln_bwd(dy.contiguous(), x, weight, mean, rstd, None, ctx.with_bias, dw, db, acc)
        if acc:
            return dx, None, None
        return dx, dw, db


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, *params):
        need = _grad_mode() and any(ctx.needs_input_grad)
        out, saved = ops.mdta_fwd(x, None, params, heads, need)
        if need:
            ctx.heads = heads
            ctx.mg = _main_grads(params)
            ctx.n_params = len(params)
            ctx.save_for_backward(x, saved, *[p for p in params if p is not None])
            ctx.present = [p is not None for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        x, saved, *rest = ctx.saved_tensors
        it = iter(rest)
        params = tuple(next(it) if pr else None for pr in ctx.present)
        acc = ctx.mg is not None
        grads = ctx.mg if acc else _fresh_grads(params)
        dx = ops.mdta_bwd(x, dout.contiguous(), params, ctx.heads, saved, grads, acc)
        return (dx, None) + tuple(None if acc else g for g in grads)


class _FeedForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        need = _grad_mode() and any(ctx.needs_input_grad)
        out, saved = ops.gdfn_fwd(x, None, params, need)
        if need:
            ctx.mg = _main_grads(params)
            ctx.save_for_backward(x, saved, *[p for p in params if p is not None])
            ctx.present = [p is not None for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        x, saved, *rest = ctx.saved_tensors
        it = iter(rest)
        params = tuple(next(it) if pr else None for pr in ctx.present)
        acc = ctx.mg is not None
        grads = ctx.mg if acc else _fresh_grads(params)
        dx = ops.gdfn_bwd(x, dout.contiguous(), params, saved, grads, acc)
        return (dx,) + tuple(None if acc else g for g in grads)


def _block_plan(x: Tensor, heads: int, params, need: bool) -> dict:
    """Which fused forms a block's two halves take (shape / dtype / switches only: also what the custom op's fake
    implementation reads).  Half-blocks whose backward can end in the one-launch tail (weight gradient + W^T dY + LayerNorm
    backward + residual add, csrc/bwd_tail.hip) rebuild LN(x) from x and the statistics, so its output is not kept - and where
    the first 1x1 conv can normalise its input as it loads it (mi_*_fwd_ln) it is never written at all."""
    n1, att, ffn = params[0:2], params[2:9], params[11:17]
    wb = n1[1] is not None
    ks_a, hidden, ks_f = att[3].shape[-1], ffn[4].shape[1], ffn[2].shape[-1]
    tail_ok = need and wb and not ops.env("MI_NO_BWD_TAIL")                                # (A/B switches)
    tail_a = bool(tail_ok and ops.mdta_bwd_ln_ok(x, heads, ks_a, att[2] is not None))
    tail_f = bool(tail_ok and ops.gdfn_bwd_ln_ok(x, hidden, ks_f, ffn[1] is not None))
    head_ok = not ops.env("MI_NO_LN_HEAD")
    # training forward of the second half-block in ONE launch (csrc/fused_gdfn.hip, SAVE form; needs the tail in backward - LN(y)
    # is never written - and a covered plane).  OPT-IN (MI_FUSED_TRAIN=1): measured 0.85-0.99x of the chain at bs 32
    # (profiles/r03_j_*: the chain streams at 4.4-5.1 TB/s, the one-launch kernel is issue-bound at 2 waves per SIMD and
    # its 8 C planes of stores do not hide behind it)
    fused_f = bool(need and tail_f and ops.env("MI_FUSED_TRAIN") and ops.gdfn_fused_train_ok(x, hidden, ks_f))
    return {"wb": wb, "tail_a": tail_a, "tail_f": tail_f, "fused_f": fused_f,
            "head_a": bool(head_ok and (tail_a or not need) and ops.mdta_fwd_ln_ok(x, heads, ks_a)),
            "head_f": bool(head_ok and (tail_f or not need) and ops.gdfn_fwd_ln_ok(x, hidden, ks_f))}


def hidden_of(ffn) -> int:
    return ffn[4].shape[1]


def _block_forward(x: Tensor, heads: int, params, need: bool):
    """x + attn(norm1(x)) ; + ffn(norm2(.))   (Restormer.py:146-150) with both residual adds fused into the producing 1x1 GEMM
    epilogues.  -> (out, saved): ``saved`` = [xn, y, yn, mean1, rstd1, mean2, rstd2, sv_a, sv_f] (entries None where the
    backward does not need them; all None when ``need`` is False).  Shared by the autograd.Function (_BlockFn) and the
    torch.library custom op (torch_ops.transformer_block_fwd)."""
    n1, att, n2, ffn = params[0:2], params[2:9], params[9:11], params[11:17]
    plan = _block_plan(x, heads, params, need)
    wb = plan["wb"]
    xn = yn = None
    if plan["head_a"]:
        y, sv_a, mean1, rstd1 = ops.mdta_fwd(x, x, att, heads, need, ln=(n1[0], n1[1], need))
    else:
        xn, mean1, rstd1 = ops.ln_fwd(x, n1[0], n1[1], wb, want_stats=need)
        y, sv_a = ops.mdta_fwd(xn, x, att, heads, need)
    if plan["fused_f"]:
        # (packed per call: the weights change every step; one small launch)
        out, sv_f, mean2, rstd2 = ops.gdfn_fused_fwd_train(y, ops.gdfn_fused_pack(y, n2[0], n2[1], tuple(ffn)), hidden_of(ffn), wb)
    elif plan["head_f"]:
        out, sv_f, mean2, rstd2 = ops.gdfn_fwd(y, y, ffn, need, ln=(n2[0], n2[1], need))
    else:
        yn, mean2, rstd2 = ops.ln_fwd(y, n2[0], n2[1], wb, want_stats=need)
        out, sv_f = ops.gdfn_fwd(yn, y, ffn, need)
    if not need:
        return out, [None] * 9
    # (a half-block on the tail keeps no LayerNorm output: that IS the flag the backward reads)
    return out, [None if plan["tail_a"] else xn, y, None if plan["tail_f"] else yn, mean1, rstd1, mean2, rstd2, sv_a, sv_f]


def _block_backward(x: Tensor, saved, dout: Tensor, heads: int, params, grads, acc: bool) -> Tensor:
    """Backward of _block_forward: dx; parameter gradients are written (acc: accumulated) into ``grads``."""
    xn, y, yn, mean1, rstd1, mean2, rstd2, sv_a, sv_f = saved
    n1, att, n2, ffn = params[0:2], params[2:9], params[9:11], params[11:17]
    g1, ga, g2, gf = grads[0:2], grads[2:9], grads[9:11], grads[11:17]
    wb = n1[1] is not None
    dout = dout.contiguous()
    if yn is None:                     # one-launch tail: weight gradient + W^T dY + LayerNorm backward + residual
        dy = ops.gdfn_bwd(y, dout, ffn, sv_f, gf, acc, ln=(n2[0], n2[1], mean2, rstd2, dout, g2[0], g2[1]))
    else:
        dyn = ops.gdfn_bwd(yn, dout, ffn, sv_f, gf, acc)
        dy = ops.ln_bwd(dyn, y, n2[0], mean2, rstd2, dout, wb, g2[0], g2[1], acc)
    if xn is None:
        return ops.mdta_bwd(x, dy, att, heads, sv_a, ga, acc, ln=(n1[0], n1[1], mean1, rstd1, dy, g1[0], g1[1]))
    dxn = ops.mdta_bwd(xn, dy, att, heads, sv_a, ga, acc)
    return ops.ln_bwd(dxn, x, n1[0], mean1, rstd1, dy, wb, g1[0], g1[1], acc)


class _BlockFn(torch.autograd.Function):
    """The whole TransformerBlock as one autograd node (the implementation under torch_ops.transformer_block, and the direct
    route with MI_TORCH_OPS=0)."""

    N_LN, N_ATT, N_FFN = 2, 7, 6

    @staticmethod
    def forward(ctx, x, heads, *params):
        need = _grad_mode() and any(ctx.needs_input_grad)
        out, saved = _block_forward(x, heads, params, need)
        if need:
            ctx.heads = heads
            ctx.mg = _main_grads(params)
            ctx.present = [p is not None for p in params]
            ctx.save_for_backward(x, *saved, *[p for p in params if p is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        x, *rest = ctx.saved_tensors
        saved, rest = rest[:9], rest[9:]
        it = iter(rest)
        params = tuple(next(it) if pr else None for pr in ctx.present)
        acc = ctx.mg is not None
        grads = ctx.mg if acc else _fresh_grads(params)
        dx = _block_backward(x, saved, dout, ctx.heads, params, grads, acc)
        return (dx, None) + tuple(None if acc else g for g in grads)


def _fused_gdfn_pack(holder, like: Tensor, ln_params, ffn_params) -> Tensor:
    """Packed weight images of the one-launch LN + GDFN kernel, cached on the module.  The key holds every way the eight
    parameters can change: their storage (replaced parameters), their version counters (every in-place write torch makes:
    ``load_state_dict``, ``copy_``, torch optimizers) and ``ops.weights_epoch()`` - the counter the raw-pointer writers of this
    package bump (``FlatTrainer.optimizer_step`` / ``weights_changed`` / ``load_state_dict``, ``PackedWeights.refresh``): the fused
    AdamW kernel updates the flat parameter buffer without touching any version counter (a train -> validate ->
    train -> validate loop ran every later validation's second half-block on the first validation's weights)."""
    ps = tuple(ln_params) + tuple(ffn_params)
    key = (tuple((p.data_ptr(), p._version) if p is not None else None for p in ps) + (like.shape[2], like.shape[3])
           + (ops.weights_epoch(),))
    cache = getattr(holder, "_fg_pack", None)
    if cache is None or cache[0] != key:
        cache = (key, ops.gdfn_fused_pack(like, ln_params[0], ln_params[1], tuple(ffn_params)))
        holder._fg_pack = cache
    return cache[1]


def _fused_mdta_pack(holder, like: Tensor, heads: int, ln_params, att_params) -> Tensor:
    """Packed weights of the one-launch LN -> qkv -> dw3x3 -> q k^T kernel (csrc/fused_mdta.hip), cached like _fused_gdfn_pack."""
    ps = tuple(ln_params) + tuple(att_params[1:5])
    key = (tuple((p.data_ptr(), p._version) if p is not None else None for p in ps) + (ops.weights_epoch(),))
    cache = getattr(holder, "_fm_pack", None)
    if cache is None or cache[0] != key:
        cache = (key, ops.mdta_fused_pack(like, heads, ln_params[0], ln_params[1], tuple(att_params)))
        holder._fm_pack = cache
    return cache[1]


# ---- fp8 (e4m3) MFMA operands in the 1x1 projections: the tiled-inference configuration (BASELINE configs[4]) -------------------
# Activations stay bf16 in HBM; a projection's input and weight are divided by a power-of-two scale and rounded to e4m3 in
# registers on their way into v_mfma_f32_16x16x32_fp8_fp8 (csrc/pw_gemm.hip, PwwOp).  Weight scales come from the weights
# (max |W|; for MDTA's per-image project_out . softmax product the largest per-head absolute row sum of project_out, which
# bounds it).  Activation scales are static: ``fp8_calibrate`` runs the bf16 network once over sample tiles, every block records
# the largest magnitude its two projections saw, and the scale leaves F8_HEADROOM x room above it (e4m3 keeps 3 mantissa bits
# over ~15 binades, so headroom costs no precision; values past 448 x scale saturate).
F8_HEADROOM = 4.0
F8_MAX = 448.0
F8_COUNTS = {"f8": 0, "bf16": 0}       # projections run in each form since the last reset (coverage report of fp8 mode)


def _f8_pow2(bound: float) -> float:
    return 2.0 ** math.ceil(math.log2(max(float(bound), 1e-30) / F8_MAX))


def _blocks(model) -> List["TransformerBlock"]:
    return [m for m in model.modules() if isinstance(m, TransformerBlock)]


def fp8_calibrate(model, samples: Sequence[Tensor]) -> None:
    """Record activation ranges for the fp8 projections: runs ``model`` (bf16, no_grad) over ``samples`` with every
    TransformerBlock on its unfused path, then stores the four scales of each half-block on the block.  Re-run after the
    weights change."""
    blocks = _blocks(model)
    dev = next(model.parameters()).device
    for b in blocks:
        b._f8_cal = torch.zeros(4, dtype=torch.float32, device=dev)     # amax of: norm1 out, v, norm2 out, gated hidden
    try:
        with torch.no_grad():
            for smp in samples:
                model(smp)
    finally:
        cal = [b.__dict__.pop("_f8_cal") for b in blocks]
    amax = torch.stack(cal).tolist()                                    # one read-back for the whole network
    for b, a in zip(blocks, amax):
        wq, wo = b.attn.qkv.weight, b.attn.project_out.weight
        c = wo.shape[1] // b.attn.num_heads
        wo_bound = wo.detach().abs().reshape(wo.shape[0], b.attn.num_heads, c).sum(-1).max()
        wmax = torch.stack([wq.detach().abs().max(), wo_bound, b.ffn.project_in.weight.detach().abs().max(),
                            b.ffn.project_out.weight.detach().abs().max()]).tolist()
        b._f8 = {"attn": (_f8_pow2(F8_HEADROOM * a[0]), _f8_pow2(wmax[0]), _f8_pow2(F8_HEADROOM * a[1]), _f8_pow2(wmax[1])),
                 "ffn": (_f8_pow2(F8_HEADROOM * a[2]), _f8_pow2(wmax[2]), _f8_pow2(F8_HEADROOM * a[3]), _f8_pow2(wmax[3]))}
        if b.norm2.body.__class__.__name__ == "WithBias_LayerNorm":
            # the one-launch kernel multiplies the NORMALISED input ((y - mu) rstd: at most sqrt(C) in magnitude, a hard bound)
            # with W_in . diag(gamma)
            win = b.ffn.project_in.weight.detach()
            wfold = float((win.reshape(win.shape[0], -1) * b.norm2.body.weight.detach()[None, :]).abs().max())
            b._f8["ffn_fused"] = (_f8_pow2(math.sqrt(win.shape[1])), _f8_pow2(wfold), b._f8["ffn"][2], b._f8["ffn"][3])
        b._f8_key = _f8_weights_key(b)


def _f8_weights_key(block):
    """What the fp8 scales of a block were derived from: the parameters' version counters and the raw-pointer update epoch."""
    return (ops.weights_epoch(), sum(p._version for p in block.parameters()))


def fp8_projections(model, mode: Optional[str]) -> None:
    """Switch the no_grad forward of every TransformerBlock: ``"all"`` = all four 1x1 projections on fp8 MFMA operands,
    ``"attn"`` = qkv and project_out of the attention half only (the feed-forward half stays on the one-launch bf16 kernel
    where that covers the shape), ``None`` = bf16.  Needs ``fp8_calibrate`` first."""
    if mode not in (None, "all", "attn"):
        raise ValueError("fp8_projections: mode is None, 'all' or 'attn'")
    for b in _blocks(model):
        if mode is not None and getattr(b, "_f8", None) is None:
            raise RuntimeError("fp8_projections: run fp8_calibrate(model, samples) first")
        b._f8_mode = mode


def _block_calibrate(block, x: Tensor, params, cal: Tensor) -> Tensor:
    """The block's bf16 forward, unfused, recording the magnitudes the four projections read."""
    n1, att, n2, ffn = params[0:2], params[2:9], params[9:11], params[11:17]
    wb = n1[1] is not None
    Cc = x.shape[1]
    xn, _, _ = ops.ln_fwd(x, n1[0], n1[1], wb, want_stats=False)
    v = ops.dwconv_fwd(ops.conv1x1(xn, att[1][2 * Cc:], None if att[2] is None else att[2][2 * Cc:]),
                       att[3][2 * Cc:].contiguous(), None if att[4] is None else att[4][2 * Cc:].contiguous())
    y, _ = ops.mdta_fwd(xn, x, att, block.attn.num_heads, False)
    yn, _, _ = ops.ln_fwd(y, n2[0], n2[1], wb, want_stats=False)
    g = ops.dwconv_gate_fwd(ops.conv1x1(yn, ffn[0], ffn[1]), ffn[2], ffn[3], want_y=False)[1]
    out, _ = ops.gdfn_fwd(yn, y, ffn, False)
    seen = torch.stack([xn.abs().max(), v.abs().max(), yn.abs().max(), g.abs().max()]).float()
    torch.maximum(cal, seen, out=cal)
    return out


def _block_infer(block, x: Tensor, params) -> Tensor:
    """TransformerBlock.forward under no_grad (Restormer.py:146-150): nothing is saved, and the second half of the block
    (norm2 -> ffn -> +x) runs as ONE kernel where the fused GDFN covers the shape (bf16, C = 48 / 96, tile-aligned planes).
    With ``fp8_projections`` on, the 1x1 projections take fp8 MFMA operands where the kernel form allows."""
    _gpu_block_input(x)
    cal = getattr(block, "_f8_cal", None)
    if cal is not None:
        return _block_calibrate(block, x, params, cal)
    n1, att, n2, ffn = params[0:2], params[2:9], params[9:11], params[11:17]
    wb = n1[1] is not None
    heads = block.attn.num_heads
    mode = getattr(block, "_f8_mode", None) if x.dtype == torch.bfloat16 else None
    if mode and getattr(block, "_f8_key", None) != _f8_weights_key(block):
        raise RuntimeError("fp8 projections: the weights changed since fp8_calibrate(model, samples) derived the scales "
                           "(an optimizer step, load_state_dict or an in-place write); calibrate again or switch fp8 off "
                           "with fp8_projections(model, None)")
    ks_a = att[3].shape[-1]
    ln_a = ops.mdta_fwd_ln_ok(x, heads, ks_a) and not ops.env("MI_NO_LN_HEAD")        # norm1 inside the qkv GEMM
    f8_a = block._f8["attn"] if mode and ops.mdta_fwd_f8_ok(x, heads, ks_a, bool(ln_a)) else None
    if mode:
        F8_COUNTS["f8" if f8_a else "bf16"] += 2
    if (f8_a is None and not ops.env("MI_NO_FUSED_INFER") and ops.mdta_fused_ok(x, heads, ks_a)
            and (ops.mdta_fused_pays(x, heads, ks_a) or ops.env("MI_FUSED_MDTA_ALWAYS"))):
        # pass A in one launch: LN -> qkv -> dw3x3 -> q k^T partials; only v is written (q, k, qkv0 never reach HBM).  Taken
        # where it fills the chip (one persistent workgroup per CU: mi_mdta_fused_pays); small batches of small planes stay on
        # the chain, which is faster there (MI_FUSED_MDTA_ALWAYS=1 forces it: tests)
        y = ops.mdta_fused_fwd(x, _fused_mdta_pack(block, x, heads, n1, att), att, heads, wb, x)[0]
    elif ln_a:
        y = ops.mdta_fwd(x, x, att, heads, False, ln=(n1[0], n1[1], False), f8=f8_a)
        y = y if f8_a else y[0]
    else:
        xn, _, _ = ops.ln_fwd(x, n1[0], n1[1], wb, want_stats=False)
        y = ops.mdta_fwd(xn, x, att, heads, False, f8=f8_a)
        y = y if f8_a else y[0]
    hidden, ks = ffn[4].shape[1], ffn[2].shape[-1]
    if mode == "all":
        if (ops.gdfn_fused_ok(y, hidden, ks) and "ffn_fused" in block._f8 and not ops.env("MI_NO_FUSED_INFER")
                and not ops.env("MI_FG_CFG")):
            F8_COUNTS["f8"] += 2                                                          # the one-launch half-block on fp8 operands
            return ops.gdfn_fused_fwd(y, _fused_gdfn_pack(block, y, n2, ffn), hidden, wb, f8=block._f8["ffn_fused"])[0]
        ln_f = ops.gdfn_fwd_ln_ok(y, hidden, ks) and not ops.env("MI_NO_LN_HEAD")
        if ops.gdfn_fwd_f8_ok(y, hidden, ks, bool(ln_f)):
            F8_COUNTS["f8"] += 2
            if ln_f:
                return ops.gdfn_fwd(y, y, ffn, False, ln=(n2[0], n2[1], False), f8=block._f8["ffn"])
            yn, _, _ = ops.ln_fwd(y, n2[0], n2[1], wb, want_stats=False)
            return ops.gdfn_fwd(yn, y, ffn, False, f8=block._f8["ffn"])
    if mode:
        F8_COUNTS["bf16"] += 2
    if ops.gdfn_fused_ok(y, hidden, ks) and not ops.env("MI_NO_FUSED_INFER"):      # (A/B switch)
        pack = _fused_gdfn_pack(block, y, n2, ffn)
        out, _, _ = ops.gdfn_fused_fwd(y, pack, hidden, wb, want_stats=False)
        return out
    yn, _, _ = ops.ln_fwd(y, n2[0], n2[1], wb, want_stats=False)
    out, _ = ops.gdfn_fwd(yn, y, ffn, False)
    return out


def _gpu_block_input(x: Tensor) -> None:
    if not x.is_contiguous():
        raise RuntimeError("image_restoration_amd ops need contiguous NCHW tensors")


class _CrossAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, heads, *params):
        need = _grad_mode() and any(ctx.needs_input_grad)
        out, saved = ops.xmdta_fwd(x, y, None, params, heads, need)
        if need:
            ctx.heads = heads
            ctx.mg = _main_grads(params)
            ctx.present = [p is not None for p in params]
            ctx.save_for_backward(x, y, saved, *[p for p in params if p is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, saved, *rest = ctx.saved_tensors
        it = iter(rest)
        params = tuple(next(it) if pr else None for pr in ctx.present)
        acc = ctx.mg is not None
        grads = ctx.mg if acc else _fresh_grads(params)
        dx, dy = ops.xmdta_bwd(x, y, dout.contiguous(), params, ctx.heads, saved, grads, acc)
        return (dx, dy, None) + tuple(None if acc else g for g in grads)


class _DwConvFn(torch.autograd.Function):
    """Stand-alone depthwise k x k conv (stride 1, pad k/2): moce_ir.py:377-381 FFTAttention q_dwconv / kv_dwconv."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = ops.dwconv_fwd(x, weight, bias)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
            ctx.mg = _main_grads((weight, bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = ops.dwconv_bwd(dy.contiguous(), x, weight, ctx.has_bias, ctx.mg)      # (main_grad: accumulated in place)
        if ctx.mg is not None:
            return dx, None, None
        return dx, dw, db


class _Conv1x1Fn(torch.autograd.Function):
    """1x1 conv over an optional channel concat [x1 ; x2] without materialising the concat
    (Restormer.py:223,228 reduce_chan_level{3,2} applied to torch.cat, :259-266)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias):
        y = ops.conv1x1(x1, weight, bias, None, False, x2)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x1, x2, weight)
            ctx.has_bias = bias is not None
            ctx.mg = _main_grads((weight, bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, weight = ctx.saved_tensors
        dy = dy.contiguous()
        k1 = x1.shape[1]
        w2 = weight.reshape(weight.shape[0], -1)
        # the two panels' weights are column blocks of one [M, K1+K2] matrix, used in place (row stride K1+K2)
        dx1 = _dgrad_panel(dy, w2, 0, k1)
        dx2 = _dgrad_panel(dy, w2, k1, w2.shape[1] - k1) if x2 is not None else None
        acc = ctx.mg is not None
        dw = ctx.mg[0] if acc else torch.empty_like(weight)
        _wgrad_panels(dy, x1, x2, dw.reshape(w2.shape), acc)
        db = None
        if ctx.has_bias:
            if acc:
                ops.chan_sum(dy, ctx.mg[1], True)
            else:
                db = ops.chan_sum(dy)
        return dx1, dx2, (None if acc else dw), db


def _conv1x1_module(x1: Tensor, x2: Optional[Tensor], conv: nn.Conv2d) -> Tensor:
    """``conv(torch.cat([x1, x2], 1))`` for a 1x1 conv module, concat-free, with the module's forward hooks run."""
    y = _apply(_Conv1x1Fn, x1, x2, conv.weight, conv.bias)
    _fire_forward_hooks(conv, (x1, x2), y)
    return y


def _dgrad_panel(dy: Tensor, w2: Tensor, k0: int, k: int) -> Tensor:
    """dx[:, k0:k0+k] = W[:, k0:k0+k]^T dy  using the column block in place (row stride stays K1+K2)."""
    import ctypes as C
    from . import _lib as L
    B, M, H, W = dy.shape
    N = H * W
    dx = torch.empty((B, k, H, W), dtype=dy.dtype, device=dy.device)
    d = L.PwDesc()
    d.x1, d.x1_bs, d.k1 = dy.data_ptr(), M * N, M
    d.w = w2.data_ptr() + 4 * k0
    d.w_sm, d.w_sk = 1, w2.shape[1]
    d.y, d.y_bs = dx.data_ptr(), k * N
    d.m, d.n, d.batch, d.groups, d.dtype = k, N, B, 1, ops._dt(dy)
    ops.pw_gemm_desc(d, dy.device)
    return dx


def _wgrad_panels(dy: Tensor, x1: Tensor, x2: Optional[Tensor], dw2: Tensor, accumulate: bool) -> None:
    import ctypes as C
    from . import _lib as L
    B, M, H, W = dy.shape
    N = H * W
    ld = dw2.shape[1]
    k0 = 0
    for xp in (x1, x2):
        if xp is None:
            continue
        k = xp.shape[1]
        d = L.GramDesc()
        d.a, d.a_bs, d.ma = dy.data_ptr(), M * N, M
        d.b, d.b_bs, d.mb = xp.data_ptr(), k * N, k
        d.n, d.batch, d.groups, d.dtype = N, B, 1, ops._dt(dy)
        d.sum_batch, d.accumulate = 1, int(accumulate)
        d.out, d.out_ld, d.out_zs = dw2.data_ptr() + 4 * k0, ld, 0
        ws = ops._blob(L.lib().mi_gram_workspace(C.byref(d)), dy.device)
        L.check(L.lib().mi_gram(C.byref(d), ws.data_ptr(), ops._stream()), "gram(wgrad panel)")
        k0 += k


# ======================================================================================
# modules (reference interface)
# ======================================================================================
def _shape1(normalized_shape):
    if isinstance(normalized_shape, numbers.Integral):
        normalized_shape = (normalized_shape,)
    normalized_shape = torch.Size(normalized_shape)
    assert len(normalized_shape) == 1
    return normalized_shape


class BiasFree_LayerNorm(nn.Module):
    """Parameter holder for the bias-free flavour (Restormer.py:25-39)."""

    def __init__(self, normalized_shape):
        super().__init__()
        self.normalized_shape = _shape1(normalized_shape)
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))


class WithBias_LayerNorm(nn.Module):
    """Parameter holder for the with-bias flavour (Restormer.py:41-57)."""

    def __init__(self, normalized_shape):
        super().__init__()
        self.normalized_shape = _shape1(normalized_shape)
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(self.normalized_shape))


class LayerNorm(nn.Module):
    """Per-pixel LayerNorm over channels of an NCHW map (Restormer.py:60-70)."""

    def __init__(self, dim, LayerNorm_type):
        super().__init__()
        self.body = BiasFree_LayerNorm(dim) if LayerNorm_type == 'BiasFree' else WithBias_LayerNorm(dim)

    def _params(self):
        return self.body.weight, getattr(self.body, "bias", None)

    def forward(self, x):
        if _use_torch_ops(x):
            return _torch_ops().layernorm(x, *self._params())
        return _apply(_LayerNormFn, x, *self._params())


class FeedForward(nn.Module):
    """GDFN (Restormer.py:76-93)."""

    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden_features = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden_features * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=3, stride=1, padding=1,
                                groups=hidden_features * 2, bias=bias)
        self.project_out = nn.Conv2d(hidden_features, dim, kernel_size=1, bias=bias)

    def _params(self):
        return (self.project_in.weight, self.project_in.bias, self.dwconv.weight, self.dwconv.bias,
                self.project_out.weight, self.project_out.bias)

    def forward(self, x):
        if _use_torch_ops(x):
            return _torch_ops().gdfn(x, self._params())
        return _apply(_FeedForwardFn, x, *self._params())


class Attention(nn.Module):
    """MDTA (Restormer.py:99-132)."""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def _params(self):
        return (self.temperature, self.qkv.weight, self.qkv.bias, self.qkv_dwconv.weight, self.qkv_dwconv.bias,
                self.project_out.weight, self.project_out.bias)

    def forward(self, x):
        if _use_torch_ops(x):
            return _torch_ops().mdta(x, self.num_heads, self._params())
        return _apply(_AttentionFn, x, self.num_heads, *self._params())


class TransformerBlock(nn.Module):
    """norm1 -> attn -> +x -> norm2 -> ffn -> +x (Restormer.py:137-150), one fused autograd node."""

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x):
        params = self.norm1._params() + self.attn._params() + self.norm2._params() + self.ffn._params()
        if not torch.is_grad_enabled() and x.is_cuda:
            return _block_infer(self, x, params)
        return block_apply(x, self.attn.num_heads, params)


class _Conv3x3Fn(torch.autograd.Function):
    """Dense 3x3 convolution (stride 1, pad 1).  bf16 planes with W % 8 == 0 take the implicit-GEMM kernels (csrc/conv3x3.hip:
    forward, data gradient and weight gradient without the 9-plane expansion in HBM); everything else (fp32 - the exact parity
    path -, odd widths) is built from the native 1x1 GEMM / Gram plus the im2col3x3 / col2im3x3 layout kernels (csrc/glue.hip): Restormer's OverlapPatchEmbed (3 -> dim, Restormer.py:156-165), output conv (2*dim -> 3 plus the
    input residual, :243,281) and the C -> C/2 / C -> 2C convs of Downsample / Upsample (:171-189).  The form with the
    fewer expanded planes is taken (im2col when Cin <= Cout, col2im otherwise), so no vendor convolution runs in the step.
      Cin tiny : y = W[Cout,9Cin] . im2col(x);            dW = Gram(dy, im2col(x));  dx = col2im(W^T . dy)
      Cout tiny: y = col2im(Wz[9Cout,Cin] . x) (+b)(+res); dz = im2col_flipped(dy);   dx = Wz^T . dz;  dWz = Gram(dz, x)"""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        cout, cin = weight.shape[0], weight.shape[1]
        # (a 2..3-channel OUTPUT - the network's last conv - stays on the col2im form: 16-row MFMA tiles would be 80 % padding,
        #  measured 0.39x; profiles/r03_o_*)
        ctx.implicit = bool(cout >= 16 and ops.conv3x3_ok(x) and ops_dense(x) and (residual is None or ops_dense(residual))
                            and not ops.env("MI_NO_CONV3_IMPLICIT"))
        if ctx.implicit:
            # implicit GEMM (csrc/conv3x3.hip): no 9-plane expansion in HBM; backward = the same kernel on the flipped /
            # transposed weight pack (dx) + the pixel-contraction kernel (dW)
            y = ops.conv3x3(x, weight, bias, residual)
            if _grad_mode() and any(ctx.needs_input_grad):
                ctx.save_for_backward(x, weight)
                ctx.has_bias = bias is not None
                ctx.mg = _main_grads((weight, bias))
            return y
        ctx.small_in = cin <= cout
        if ctx.small_in:
            xcol = ops.im2col3x3(x)
            y = ops.conv1x1(xcol, weight.reshape(cout, cin * 9), bias, residual)
            saved = xcol
        else:
            wz = weight.permute(0, 2, 3, 1).reshape(cout * 9, cin).contiguous()      # [(co,ky,kx), ci]; a few KB
            y = ops.col2im3x3(ops.conv1x1(x, wz), bias, residual)
            saved = x
        if _grad_mode() and any(ctx.needs_input_grad):
            # The weight gradient needs the 9-plane expansion of x again.  Default: keep it from the forward (1.6 GB at bs 32 over the
            # three Upsample convs of Restormer base).  MI_CONV3_RECOL=1: the memory-lean form - for a wide input keep x only and
            # rebuild the expansion in backward (three more im2col launches per step, ~0.9 ms).
            ctx.recol = ctx.small_in and cin > 4 and bool(ops.env("MI_CONV3_RECOL"))
            ctx.save_for_backward(x if ctx.recol else saved, weight)
            ctx.has_bias = bias is not None
            ctx.mg = _main_grads((weight, bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        saved, weight = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = weight.shape[0], weight.shape[1]
        dx = None
        if ctx.implicit:
            acc = ctx.mg is not None
            dw = ops.conv3x3_wgrad(dy, saved, ctx.mg[0] if acc else None, acc)
            if ctx.needs_input_grad[0]:
                dx = ops.conv3x3(dy, weight, transpose=True)
            db = None
            if ctx.has_bias:
                db = ops.chan_sum(dy, ctx.mg[1] if acc else None, acc)
            dres = dy if ctx.needs_input_grad[3] else None
            if acc:
                return dx, None, None, dres
            return dx, dw, db, dres
        if ctx.small_in:
            if ctx.recol:
                saved = ops.im2col3x3(saved)
            dw = ops.gram(dy, saved, 1, True)[0].reshape(weight.shape)                # [Cout, 9Cin]
            if ctx.needs_input_grad[0]:
                dx = ops.col2im3x3(ops.conv1x1(dy, weight.reshape(cout, cin * 9), None, None, True), flip=True)
        else:
            dz = ops.im2col3x3(dy, flip=True)                                          # [(co,ky,kx)] planes
            dwz = ops.gram(dz, saved, 1, True)[0]                                      # [9Cout, Cin]
            dw = dwz.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
            if ctx.needs_input_grad[0]:
                wz = weight.permute(0, 2, 3, 1).reshape(cout * 9, cin).contiguous()
                dx = ops.conv1x1(dz, wz, None, None, True)
        db = None
        if ctx.has_bias:
            db = ops.chan_sum(dy, ctx.mg[1] if ctx.mg is not None else None, ctx.mg is not None)
        dres = dy if ctx.needs_input_grad[3] else None
        if ctx.mg is not None:
            ctx.mg[0].add_(dw)
            return dx, None, None, dres
        return dx, dw.contiguous(), db, dres


def _fire_forward_hooks(module: nn.Module, inputs, output) -> None:
    """Modules this package uses functionally (their weights are read, ``module(x)`` is never called: the dense 3x3 glue convs,
    ``reduce_chan_*``, ``up2_1``) still run their registered forward hooks, so observers - the trainer's per-stage gradient
    bucketing above all - see that the module took part in the step."""
    if module._forward_hooks:
        for hook in list(module._forward_hooks.values()):
            hook(module, inputs, output)


def _conv2d(x: Tensor, conv: nn.Conv2d, residual: Optional[Tensor] = None, owner: Optional[nn.Module] = None) -> Tensor:
    """U-Net glue convolution (dense 3x3, stride 1, pad 1: SURVEY 8(f) row f1) on the native kernels (_Conv3x3Fn); rows of
    16..256 pixels (power of two) take the wave-streaming layout kernels, every other plane the general form.  Anything else
    (another kernel size / stride / groups, a CPU tensor) raises: no vendor convolution runs on this path."""
    if not (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1
            and conv.dilation == (1, 1)):
        raise NotImplementedError("image_restoration_amd: only the dense 3x3 / stride 1 / pad 1 glue convolution is built "
                                  f"(got kernel {conv.kernel_size}, stride {conv.stride}, padding {conv.padding}, "
                                  f"groups {conv.groups})")
    ops._gpu(x, residual)
    y = _apply(_Conv3x3Fn, x, conv.weight, conv.bias, residual)
    _fire_forward_hooks(owner if owner is not None else conv, (x,), y)
    return y


class _PixelShuffleFn(torch.autograd.Function):
    """PixelShuffle(2) / PixelUnshuffle(2) as one native streaming pass; each is the other's backward."""

    @staticmethod
    def forward(ctx, x, unshuffle):
        ctx.unshuffle = unshuffle
        return ops.pixel_shuffle2(x, unshuffle)

    @staticmethod
    def backward(ctx, dy):
        return ops.pixel_shuffle2(dy if ops_dense(dy) else dy.contiguous(), not ctx.unshuffle), None


def ops_dense(t: Tensor) -> bool:
    """[C,H,W] blocks dense (a channel slice of a contiguous NCHW tensor qualifies)."""
    return t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3] and t.stride(1) == t.shape[2] * t.shape[3]


class _UpCatFn(torch.autograd.Function):
    """cat([PixelShuffle(2)(z), skip], 1)  (Restormer.py:185-189 Upsample body + :266 torch.cat) without an intermediate:
    the shuffle writes straight into the first half of the concatenation buffer, the skip is copied into the second; the
    backward un-shuffles the first half in place of a slice copy."""

    @staticmethod
    def forward(ctx, z, skip):
        B, c4, H, W = z.shape
        c = c4 // 4
        out = torch.empty((B, c + skip.shape[1], 2 * H, 2 * W), dtype=z.dtype, device=z.device)
        ops.pixel_shuffle2(z, False, out=out[:, :c])
        ops.copy_rows(skip, out[:, c:])
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        c = ctx.c
        dz = ops.pixel_shuffle2(dout[:, :c], True)
        dskip = torch.empty((dout.shape[0], dout.shape[1] - c) + tuple(dout.shape[2:]), dtype=dout.dtype, device=dout.device)
        ops.copy_rows(dout[:, c:], dskip)
        return dz, dskip


def _shuffle(x: Tensor, unshuffle: bool) -> Tensor:
    ops._gpu(x)
    return _apply(_PixelShuffleFn, x, unshuffle)


class OverlapPatchEmbed(nn.Module):
    """3x3 conv embedding (Restormer.py:156-165)."""

    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=3, stride=1, padding=1, bias=bias)

    def forward(self, x):
        return _conv2d(x, self.proj)


class Downsample(nn.Module):
    """3x3 conv C->C/2 + PixelUnshuffle(2) (Restormer.py:171-179)."""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat // 2, kernel_size=3, stride=1, padding=1, bias=False),
                                  nn.PixelUnshuffle(2))

    def forward(self, x):
        return _shuffle(_conv2d(x, self.body[0]), True)


class Upsample(nn.Module):
    """3x3 conv C->2C + PixelShuffle(2) (Restormer.py:181-189)."""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat * 2, kernel_size=3, stride=1, padding=1, bias=False),
                                  nn.PixelShuffle(2))

    def forward(self, x):
        return _shuffle(_conv2d(x, self.body[0]), False)


def _up_cat(up: "Upsample", x: Tensor, skip: Tensor) -> Tensor:
    """torch.cat([up(x), skip], 1) (Restormer.py:265-266) with the shuffle writing into the concatenation buffer."""
    z = _conv2d(x, up.body[0], owner=up)
    ops._gpu(skip)
    return _apply(_UpCatFn, z, skip)


def _stage(dim, heads, n, ffn, bias, ln):
    return nn.Sequential(*[TransformerBlock(dim=dim, num_heads=heads, ffn_expansion_factor=ffn, bias=bias,
                                            LayerNorm_type=ln) for _ in range(n)])


class Restormer(nn.Module):
    """The reference U-Net (Restormer.py:193-284) over the HIP-backed blocks; same constructor and state_dict."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias',
                 dual_pixel_task=False):
        super().__init__()
        f, b, ln = ffn_expansion_factor, bias, LayerNorm_type
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level1 = _stage(dim, heads[0], num_blocks[0], f, b, ln)
        self.down1_2 = Downsample(dim)
        self.encoder_level2 = _stage(int(dim * 2 ** 1), heads[1], num_blocks[1], f, b, ln)
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.encoder_level3 = _stage(int(dim * 2 ** 2), heads[2], num_blocks[2], f, b, ln)
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.latent = _stage(int(dim * 2 ** 3), heads[3], num_blocks[3], f, b, ln)
        self.up4_3 = Upsample(int(dim * 2 ** 3))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 3), int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = _stage(int(dim * 2 ** 2), heads[2], num_blocks[2], f, b, ln)
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level2 = _stage(int(dim * 2 ** 1), heads[1], num_blocks[1], f, b, ln)
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.decoder_level1 = _stage(int(dim * 2 ** 1), heads[0], num_blocks[0], f, b, ln)
        self.refinement = _stage(int(dim * 2 ** 1), heads[0], num_refinement_blocks, f, b, ln)
        self.dual_pixel_task = dual_pixel_task
        if self.dual_pixel_task:
            self.skip_conv = nn.Conv2d(dim, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)

    def forward(self, inp_img):
        inp_enc_level1 = self.patch_embed(inp_img)
        out_enc_level1 = self.encoder_level1(inp_enc_level1)
        out_enc_level2 = self.encoder_level2(self.down1_2(out_enc_level1))
        out_enc_level3 = self.encoder_level3(self.down2_3(out_enc_level2))
        latent = self.latent(self.down3_4(out_enc_level3))

        # concat-free channel reduce: two K-panels of one 1x1 GEMM (Restormer.py:259-261)
        inp_dec_level3 = _conv1x1_module(self.up4_3(latent), out_enc_level3, self.reduce_chan_level3)
        out_dec_level3 = self.decoder_level3(inp_dec_level3)
        inp_dec_level2 = _conv1x1_module(self.up3_2(out_dec_level3), out_enc_level2, self.reduce_chan_level2)
        out_dec_level2 = self.decoder_level2(inp_dec_level2)
        inp_dec_level1 = _up_cat(self.up2_1, out_dec_level2, out_enc_level1)
        out_dec_level1 = self.refinement(self.decoder_level1(inp_dec_level1))

        if self.dual_pixel_task:
            out_dec_level1 = out_dec_level1 + _conv1x1_module(inp_enc_level1, None, self.skip_conv)
            return _conv2d(out_dec_level1, self.output)
        return _conv2d(out_dec_level1, self.output, inp_img)
