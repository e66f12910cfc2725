"""``torch.library`` custom ops over the C-ABI (BASELINE north_star: "Python host code registers PyTorch-ROCm custom ops over a
thin C-ABI"; SURVEY 7.1 step 3, 8(b) "Autograd").

Four module-level entry points of the reference interface are registered in the ``mi_restore`` namespace, each as a
forward / backward pair with a fake (meta) implementation and ``register_autograd``:

    mi_restore::layernorm_fwd / _bwd           LayerNorm.forward          Restormer.py:60-70
    mi_restore::mdta_fwd / _bwd                Attention.forward          Restormer.py:99-132
    mi_restore::gdfn_fwd / _bwd                FeedForward.forward        Restormer.py:76-93
    mi_restore::transformer_block_fwd / _bwd   TransformerBlock.forward   Restormer.py:137-150 (moce_ir.py:805-834 EncoderBlock)

The forward ops return ``[out, saved...]``: what the backward needs (LayerNorm statistics, the kernels' saved-for-backward
blobs) are op OUTPUTS, as the custom-op autograd contract wants; ``register_autograd`` stores them and calls the backward op.
Under the ops sit the same helpers the ``torch.autograd.Function`` nodes of ``restormer.py`` use (``_block_forward`` /
``_block_backward``, ``ops.mdta_fwd`` ...): one implementation, two front doors
(modules route through these ops while torch.compile traces and with ``MI_TORCH_OPS=1``; eager calls take the bare
autograd.Function nodes over the same implementation: restormer._use_torch_ops).

Arguments shared by all forward ops: ``need`` - build what backward needs (the caller's grad mode; an op body always runs
with grad mode off and cannot see it); ``accumulate`` (backward ops) - parameter gradients are ADDED into each parameter's
``main_grad`` buffer (the trainer's flat gradient buffer) and the returned gradients are empty placeholders.
Absent optional parameters (bias=False) are ``None``; absent saved tensors travel as empty tensors.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.library import custom_op

from . import _lib as L
from . import ops

Tensor = torch.Tensor
NS = "mi_restore"

_T = "Tensor"
_TO = "Tensor?"


def _schema(tensors: Sequence[str], opt: Sequence[bool], tail: str) -> str:
    args = ", ".join(f"{_TO if o else _T} {n}" for n, o in zip(tensors, opt))
    return f"({args}{tail}) -> Tensor[]"


def _e(like: Tensor) -> Tensor:
    """Placeholder for an absent tensor in an op's Tensor[] output: one int8 element (no real output has that dtype; a
    zero-size tensor would do, but zero-size outputs all share the null storage and trip the alias checks of opcheck)."""
    return like.new_empty(1, dtype=torch.int8)


def _absent(t: Tensor) -> bool:
    return t.dtype == torch.int8


def _pack(saved: Sequence[Optional[Tensor]], like: Tensor) -> List[Tensor]:
    return [t if t is not None else _e(like) for t in saved]


def _unpack(saved: Sequence[Tensor]) -> List[Optional[Tensor]]:
    return [None if _absent(t) else t for t in saved]


# Hand-off of the trainer's gradient buffers around the dispatcher (the ops' tensor arguments reach the op bodies as plain
# tensors: attributes of the caller's Parameter objects do not travel).  "next": set by the module-level wrappers right before a
# forward op call, consumed by that call's setup_context; "bwd": set by the autograd backward around the backward op call.
# Backward runs on autograd's thread for the device, forward on the caller's: the two keys never race on one device.
_MAIN_GRADS: dict = {}


def _grads_for(params: Sequence[Optional[Tensor]], accumulate: bool):
    """Gradient buffers of a backward op: the parameters' main_grad buffers (accumulate) or fresh tensors."""
    if accumulate:
        mg = _MAIN_GRADS.get("bwd")
        if mg is None:
            mg = [None if p is None else getattr(p, "main_grad", None) for p in params]
        if any(p is not None and g is None for p, g in zip(params, mg)):
            raise RuntimeError("accumulate=True needs a main_grad buffer on every parameter (FlatTrainer sets them)")
        return list(mg)
    return [None if p is None else torch.empty_like(p) for p in params]


def _grad_outputs(params, grads, accumulate: bool, like: Tensor) -> List[Tensor]:
    return [_e(like) if (p is None or accumulate) else g for p, g in zip(params, grads)]


def _fake_grads(params, accumulate: bool, like: Tensor) -> List[Tensor]:
    return [_e(like) if (p is None or accumulate) else torch.empty_like(p) for p in params]


def _stats(x: Tensor) -> Tensor:
    return x.new_empty((x.shape[0], x.shape[2] * x.shape[3]), dtype=torch.float32)


def _blob(x: Tensor, nbytes: int) -> Tensor:
    return x.new_empty(max(int(nbytes), 256), dtype=torch.uint8)


def _register(name: str, fwd_schema: str, bwd_schema: str, fwd_impl, fwd_fake, bwd_impl, bwd_fake, n_params: int,
              n_lead: int, n_tail: int):
    """One forward / backward op pair.  Forward inputs: ``n_lead`` leading arguments (tensors first), ``n_params`` parameters,
    ``n_tail`` trailing scalars.  The backward op takes (dout, *forward inputs without the trailing scalars, saved[], accumulate)
    and returns [d(lead tensors)..., d(params)...]."""
    fwd = custom_op(f"{NS}::{name}_fwd", mutates_args=(), schema=fwd_schema)(fwd_impl)
    fwd.register_fake(fwd_fake)
    bwd = custom_op(f"{NS}::{name}_bwd", mutates_args=(), schema=bwd_schema)(bwd_impl)
    bwd.register_fake(bwd_fake)
    bwd_op = getattr(getattr(torch.ops, NS), f"{name}_bwd")

    def setup(ctx, inputs, output):
        lead, params = inputs[:n_lead], inputs[n_lead:n_lead + n_params]
        ctx.scalars = [v for v in lead if not isinstance(v, Tensor)]
        ctx.lead_is_tensor = [isinstance(v, Tensor) for v in lead]
        ctx.n_params_present = [p is not None for p in params]
        # The tensors autograd hands to setup_context are not the caller's Parameter objects, so the trainer's per-parameter
        # ``main_grad`` buffers (attributes of those objects) are picked up from the wrapper's hand-off (_MAIN_GRADS) instead.
        ctx.mg = _MAIN_GRADS.pop("next", None)
        # only out (output 0) is differentiable: the saved tensors are outputs because the custom-op autograd contract wants
        # them to be, and unused-gradient zeros must not be materialised for them (GB-sized memsets per block otherwise)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*output[1:])
        ctx.save_for_backward(*[v for v in lead if isinstance(v, Tensor)], *[p for p in params if p is not None], *output[1:])

    def backward(ctx, grads):
        dout = grads[0]
        if dout is None:
            return (None,) * (n_lead + n_params + n_tail)
        tens = list(ctx.saved_tensors)
        n_lt = sum(ctx.lead_is_tensor)
        n_pp = sum(ctx.n_params_present)
        lead_t, ptens, saved = tens[:n_lt], tens[n_lt:n_lt + n_pp], tens[n_lt + n_pp:]
        it = iter(ptens)
        params = [next(it) if pr else None for pr in ctx.n_params_present]
        accumulate = ctx.mg is not None
        _MAIN_GRADS["bwd"] = ctx.mg
        try:
            res = bwd_op(dout.contiguous(), *lead_t, *ctx.scalars, *params, saved, accumulate)
        finally:
            _MAIN_GRADS.pop("bwd", None)
        d_lead_t = list(res[:n_lt])
        d_par = res[n_lt:]
        out = []
        it = iter(d_lead_t)
        for is_t in ctx.lead_is_tensor:
            out.append(next(it) if is_t else None)
        for p, g in zip(params, d_par):
            out.append(None if (p is None or accumulate) else g)
        return tuple(out) + (None,) * n_tail

    fwd.register_autograd(backward, setup_context=setup)
    return getattr(getattr(torch.ops, NS), f"{name}_fwd"), bwd_op


# ------------------------------------------------------------------------------------------------ LayerNorm
_LN_P = ("weight", "bias")
_LN_O = (False, True)


def _ln_fwd(x, weight, bias, need):
    y, mean, rstd = ops.ln_fwd(x, weight, bias, bias is not None, want_stats=need)
    return [y] + _pack([mean, rstd], x)


def _ln_fwd_fake(x, weight, bias, need):
    return [torch.empty_like(x)] + ([_stats(x), _stats(x)] if need else [_e(x), _e(x)])


def _ln_bwd(dout, x, weight, bias, saved, accumulate):
    mean, rstd = saved
    params = (weight, bias)
    dw, db = _grads_for(params, accumulate)
    dx = ops.ln_bwd(dout, x, weight, mean, rstd, None, bias is not None, dw, db, accumulate)
    return [dx] + _grad_outputs(params, (dw, db), accumulate, x)


def _ln_bwd_fake(dout, x, weight, bias, saved, accumulate):
    return [torch.empty_like(x)] + _fake_grads((weight, bias), accumulate, x)


layernorm_fwd, layernorm_bwd = _register(
    "layernorm",
    _schema(("x",) + _LN_P, (False,) + _LN_O, ", bool need"),
    _schema(("dout", "x") + _LN_P, (False, False) + _LN_O, ", Tensor[] saved, bool accumulate"),
    _ln_fwd, _ln_fwd_fake, _ln_bwd, _ln_bwd_fake, n_params=2, n_lead=1, n_tail=1)

# ------------------------------------------------------------------------------------------------ MDTA
_AT_P = ("temperature", "qkv_w", "qkv_b", "dw_w", "dw_b", "proj_w", "proj_b")
_AT_O = (False, False, True, False, True, False, True)


def _mdta_saved_bytes(x: Tensor, heads: int, ks: int) -> int:
    B, Cc, H, W = x.shape
    import ctypes as C
    s = L.MdtaShape(B, Cc, heads, H, W, L.MI_BF16 if x.dtype == torch.bfloat16 else L.MI_F32, ks)
    return int(L.lib().mi_mdta_saved_bytes(C.byref(s)))


def _gdfn_saved_bytes(x: Tensor, hidden: int, ks: int) -> int:
    B, Cc, H, W = x.shape
    import ctypes as C
    s = L.GdfnShape(B, Cc, hidden, H, W, L.MI_BF16 if x.dtype == torch.bfloat16 else L.MI_F32, ks,
                    1 if ops.env("MI_GDFN_STORE_Y") else 0)
    return int(L.lib().mi_gdfn_saved_bytes(C.byref(s)))


def _mdta_fwd(x, heads, *rest):
    params, need = rest[:7], rest[7]
    out, saved = ops.mdta_fwd(x, None, params, heads, need)
    return [out] + _pack([saved], x)


def _mdta_fwd_fake(x, heads, *rest):
    params, need = rest[:7], rest[7]
    return [torch.empty_like(x), _blob(x, _mdta_saved_bytes(x, heads, params[3].shape[-1])) if need else _e(x)]


def _mdta_bwd(dout, x, heads, *rest):
    params, saved, accumulate = rest[:7], rest[7], rest[8]
    grads = _grads_for(params, accumulate)
    dx = ops.mdta_bwd(x, dout, params, heads, saved[0], grads, accumulate)
    return [dx] + _grad_outputs(params, grads, accumulate, x)


def _mdta_bwd_fake(dout, x, heads, *rest):
    params, accumulate = rest[:7], rest[8]
    return [torch.empty_like(x)] + _fake_grads(params, accumulate, x)


mdta_fwd, mdta_bwd = _register(
    "mdta",
    "(Tensor x, int heads, " + _schema(_AT_P, _AT_O, ", bool need")[1:],
    "(Tensor dout, Tensor x, int heads, " + _schema(_AT_P, _AT_O, ", Tensor[] saved, bool accumulate")[1:],
    _mdta_fwd, _mdta_fwd_fake, _mdta_bwd, _mdta_bwd_fake, n_params=7, n_lead=2, n_tail=1)

# ------------------------------------------------------------------------------------------------ GDFN
_FF_P = ("in_w", "in_b", "dw_w", "dw_b", "out_w", "out_b")
_FF_O = (False, True, False, True, False, True)


def _gdfn_fwd(x, *rest):
    params, need = rest[:6], rest[6]
    out, saved = ops.gdfn_fwd(x, None, params, need)
    return [out] + _pack([saved], x)


def _gdfn_fwd_fake(x, *rest):
    params, need = rest[:6], rest[6]
    return [torch.empty_like(x), _blob(x, _gdfn_saved_bytes(x, params[4].shape[1], params[2].shape[-1])) if need else _e(x)]


def _gdfn_bwd(dout, x, *rest):
    params, saved, accumulate = rest[:6], rest[6], rest[7]
    grads = _grads_for(params, accumulate)
    dx = ops.gdfn_bwd(x, dout, params, saved[0], grads, accumulate)
    return [dx] + _grad_outputs(params, grads, accumulate, x)


def _gdfn_bwd_fake(dout, x, *rest):
    params, accumulate = rest[:6], rest[7]
    return [torch.empty_like(x)] + _fake_grads(params, accumulate, x)


gdfn_fwd, gdfn_bwd = _register(
    "gdfn",
    _schema(("x",) + _FF_P, (False,) + _FF_O, ", bool need"),
    _schema(("dout", "x") + _FF_P, (False, False) + _FF_O, ", Tensor[] saved, bool accumulate"),
    _gdfn_fwd, _gdfn_fwd_fake, _gdfn_bwd, _gdfn_bwd_fake, n_params=6, n_lead=1, n_tail=1)

# ------------------------------------------------------------------------------------------------ TransformerBlock
_BK_P = (("n1_w", "n1_b", "temperature", "qkv_w", "qkv_b", "qkv_dw_w", "qkv_dw_b", "proj_w", "proj_b", "n2_w", "n2_b")
         + ("in_w", "in_b", "ffn_dw_w", "ffn_dw_b", "out_w", "out_b"))
_BK_O = ((False, True) + _AT_O + (False, True) + _FF_O)


def _block_fwd(x, heads, *rest):
    from . import restormer as R
    params, need = rest[:17], rest[17]
    out, saved = R._block_forward(x, heads, params, need)
    return [out] + _pack(saved, x)


def _block_fwd_fake(x, heads, *rest):
    from . import restormer as R
    params, need = rest[:17], rest[17]
    out = torch.empty_like(x)
    if not need:
        return [out] + [_e(x) for _ in range(9)]
    plan = R._block_plan(x, heads, params, need)
    att, ffn = params[2:9], params[11:17]
    return [out,
            _e(x) if plan["tail_a"] else torch.empty_like(x),                  # xn
            torch.empty_like(x),                                               # y
            _e(x) if plan["tail_f"] else torch.empty_like(x),                  # yn
            _stats(x), _stats(x), _stats(x), _stats(x),
            _blob(x, _mdta_saved_bytes(x, heads, att[3].shape[-1])),
            _blob(x, _gdfn_saved_bytes(x, ffn[4].shape[1], ffn[2].shape[-1]))]


def _block_bwd(dout, x, heads, *rest):
    from . import restormer as R
    params, saved, accumulate = rest[:17], rest[17], rest[18]
    grads = _grads_for(params, accumulate)
    dx = R._block_backward(x, _unpack(saved), dout, heads, params, grads, accumulate)
    return [dx] + _grad_outputs(params, grads, accumulate, x)


def _block_bwd_fake(dout, x, heads, *rest):
    params, accumulate = rest[:17], rest[18]
    return [torch.empty_like(x)] + _fake_grads(params, accumulate, x)


transformer_block_fwd, transformer_block_bwd = _register(
    "transformer_block",
    "(Tensor x, int heads, " + _schema(_BK_P, _BK_O, ", bool need")[1:],
    "(Tensor dout, Tensor x, int heads, " + _schema(_BK_P, _BK_O, ", Tensor[] saved, bool accumulate")[1:],
    _block_fwd, _block_fwd_fake, _block_bwd, _block_bwd_fake, n_params=17, n_lead=2, n_tail=1)

OPS = {"layernorm": (layernorm_fwd, layernorm_bwd), "mdta": (mdta_fwd, mdta_bwd), "gdfn": (gdfn_fwd, gdfn_bwd),
       "transformer_block": (transformer_block_fwd, transformer_block_bwd)}


def _need(x: Tensor, params) -> bool:
    need = torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params))
    _MAIN_GRADS.pop("next", None)
    # Under torch.compile the hand-off below would run at TRACE time and the in-place accumulation into main_grad would be a
    # side effect the graph cannot see (the bwd ops declare mutates_args=()): compiled steps return their parameter
    # gradients as outputs instead (accumulate=False; FlatTrainer._fold_autograd_grads adds .grad into main_grad).
    if need and not torch.compiler.is_compiling():
        mg = [None if p is None else getattr(p, "main_grad", None) for p in params]
        if any(p is not None for p in params) and all(g is not None for p, g in zip(params, mg) if p is not None):
            _MAIN_GRADS["next"] = mg
    return need


def layernorm(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    return layernorm_fwd(x, weight, bias, _need(x, (weight, bias)))[0]


def mdta(x: Tensor, heads: int, params) -> Tensor:
    return mdta_fwd(x, heads, *params, _need(x, params))[0]


def gdfn(x: Tensor, params) -> Tensor:
    return gdfn_fwd(x, *params, _need(x, params))[0]


def transformer_block(x: Tensor, heads: int, params) -> Tensor:
    return transformer_block_fwd(x, heads, *params, _need(x, params))[0]
