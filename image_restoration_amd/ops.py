"""Tensor-level wrappers over the C-ABI (``include/mi_restore.h``).

PyTorch is plumbing here: it owns device memory (outputs, saved blobs and workspaces are torch
tensors, so the caching allocator and stream semantics apply) and supplies the current HIP stream.
All compute happens in ``libmi_restore.so``.  CPU tensors are rejected: there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


def _dt(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return L.MI_F32
    if t.dtype == torch.bfloat16:
        return L.MI_BF16
    raise TypeError(f"activations must be float32 or bfloat16, got {t.dtype}")


def _gpu(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("image_restoration_amd ops run on the MI355X only (got a CPU tensor); "
                               "the CPU restatement lives in oracle/ and is test infrastructure")
        if not t.is_contiguous():
            raise RuntimeError("image_restoration_amd ops need contiguous NCHW tensors")


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32(t: Optional[Tensor], what: str) -> Optional[Tensor]:
    if t is not None and t.dtype != torch.float32:
        raise TypeError(f"{what} must be float32 (parameters and their gradients stay fp32), got {t.dtype}")
    return t


def _stream() -> int:
    # the raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object: 9 us per call)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _blob(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- A/B switches (environment variables)
# Read once per process (a TransformerBlock forward used to make eight os.environ look-ups); ``reload_env()`` re-reads them here
# and in the library (mi_env_reload) - for tests and A/B tools that flip a switch inside one process.
_ENV: dict = {}
_WEIGHTS_EPOCH = [0]


def env(name: str) -> Optional[str]:
    """Value of an MI_* switch as the process had it at first use / the last reload_env(); None when unset or empty."""
    try:
        return _ENV[name]
    except KeyError:
        v = os.environ.get(name) or None
        _ENV[name] = v
        return v


def reload_env() -> None:
    global _WS_CACHE_ON
    _ENV.clear()
    _WS_CACHE_ON = os.environ.get("MI_WS_CACHE", "1") != "0"
    if L._lib is not None:
        L.check(L.lib().mi_env_reload(), "env_reload")


def weights_epoch() -> int:
    """Counter of parameter updates torch cannot see (the fused AdamW kernel writes through raw pointers and bumps no version
    counter): every cache of data derived from the weights keys on it (restormer._fused_gdfn_pack, the fp8 scales)."""
    return _WEIGHTS_EPOCH[0]


def bump_weights_epoch() -> None:
    _WEIGHTS_EPOCH[0] += 1


_WS_CACHE: dict = {}
_WS_CACHE_ON = os.environ.get("MI_WS_CACHE", "1") != "0"


def _ws(nbytes: int, device) -> Tensor:
    """Scratch for ONE library call.  Calls are stream-ordered and a workspace is dead when its call returns, so one grow-only
    buffer per device serves them all (a launch-bound step made ~2000 allocator calls for these; MI_WS_CACHE=0: per-call blobs)."""
    n = max(int(nbytes), 256)
    if not _WS_CACHE_ON:
        return torch.empty(n, dtype=torch.uint8, device=device)
    key = device.index
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < n:
        buf = None
        _WS_CACHE.pop(key, None)                       # release the old one first
        buf = _WS_CACHE[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return buf


# ----------------------------------------------------------------------------- LayerNorm
def ln_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], with_bias: bool, want_stats: bool = True):
    _gpu(x, w, b)
    _f32(w, "LayerNorm weight"); _f32(b, "LayerNorm bias")
    B, Cc, H, W = x.shape
    y = torch.empty_like(x)
    mean = rstd = None
    if want_stats:
        mean = torch.empty((B, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    L.check(L.lib().mi_ln_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), B, Cc, H * W, int(with_bias), _dt(x),
                              _stream()), "ln_fwd")
    return y, mean, rstd


def ln_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor], with_bias: bool,
           dw: Tensor, db: Optional[Tensor], accumulate: bool) -> Tensor:
    _gpu(dy, x, w, mean, rstd, dres, dw, db)
    B, Cc, H, W = x.shape
    dx = torch.empty_like(x)
    ws = _ws(L.lib().mi_ln_bwd_workspace(B, Cc, H * W), x.device)
    L.check(L.lib().mi_ln_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dw), _p(db), B, Cc, H * W,
                              int(with_bias), int(accumulate), _dt(x), _p(ws), _stream()), "ln_bwd")
    return dx


# ----------------------------------------------------------------------------- depthwise conv
def dwconv_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    _gpu(x, w, bias)
    B, Cc, H, W = x.shape
    ks = w.shape[-1]
    y = torch.empty_like(x)
    L.check(L.lib().mi_dwconv_fwd(_p(x), _p(w), _p(bias), _p(y), B, Cc, H, W, ks, _dt(x), _stream()), "dwconv_fwd")
    return y


def dwconv_gate_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], want_y: bool = True):
    _gpu(x, w, bias)
    B, C2, H, W = x.shape
    ks = w.shape[-1]
    y = torch.empty_like(x) if want_y else None
    g = torch.empty((B, C2 // 2, H, W), dtype=x.dtype, device=x.device)
    L.check(L.lib().mi_dwconv_gate_fwd(_p(x), _p(w), _p(bias), _p(y), _p(g), B, C2, H, W, ks, _dt(x), _stream()),
            "dwconv_gate_fwd")
    return y, g


def dwconv_bwd(dy: Tensor, x: Tensor, w: Tensor, has_bias: bool, grads: Optional[Sequence[Optional[Tensor]]] = None):
    """grads = (dw, db) fp32 buffers to ACCUMULATE into (a trainer's main_grad slots); None: fresh tensors are returned."""
    _gpu(dy, x, w)
    B, Cc, H, W = x.shape
    ks = w.shape[-1]
    dx = torch.empty_like(x)
    acc = grads is not None
    dw = _f32(grads[0], "depthwise weight gradient") if acc else torch.empty_like(w)
    db = (_f32(grads[1], "depthwise bias gradient") if acc else torch.empty(Cc, dtype=torch.float32, device=x.device)) if has_bias else None
    ws = _ws(L.lib().mi_dwconv_bwd_workspace(B, Cc, H, W, ks), x.device)
    L.check(L.lib().mi_dwconv_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, Cc, H, W, ks, 1 if acc else 0, _dt(x), _p(ws),
                                  _stream()), "dwconv_bwd")
    return dx, dw, db


def dwconv_gate_bwd(dg: Tensor, y: Tensor, x: Tensor, w: Tensor, has_bias: bool):
    _gpu(dg, y, x, w)
    B, C2, H, W = x.shape
    ks = w.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    db = torch.empty(C2, dtype=torch.float32, device=x.device) if has_bias else None
    ws = _ws(L.lib().mi_dwconv_bwd_workspace(B, C2, H, W, ks), x.device)
    L.check(L.lib().mi_dwconv_gate_bwd(_p(dg), _p(y), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, C2, H, W, ks, 0, _dt(x),
                                       _p(ws), _stream()), "dwconv_gate_bwd")
    return dx, dw, db


# ----------------------------------------------------------------------------- pointwise GEMM / Gram
def dwconv_gate_recompute_ok(H: int, W: int, ks: int) -> bool:
    return bool(L.lib().mi_dwconv_gate_recompute_ok(H, W, ks))


def dwconv_gate_bwd_recompute(dg: Tensor, x: Tensor, w: Tensor, bias: Optional[Tensor]):
    """Gate backward with the conv outputs recomputed from the conv input x (no stored y)."""
    _gpu(dg, x, w, bias)
    B, C2, H, W = x.shape
    ks = w.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    db = torch.empty(C2, dtype=torch.float32, device=x.device) if bias is not None else None
    ws = _ws(L.lib().mi_dwconv_bwd_workspace(B, C2, H, W, ks), x.device)
    L.check(L.lib().mi_dwconv_gate_bwd_recompute(_p(dg), _p(x), _p(w), _p(bias), _p(dx), _p(dw), _p(db), B, C2, H, W, ks, 0,
                                                 _dt(x), _p(ws), _stream()), "dwconv_gate_bwd_recompute")
    return dx, dw, db


def conv1x1(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
            transposed: bool = False, x2: Optional[Tensor] = None, out: Optional[Tensor] = None,
            f8: Optional[Tuple[float, float]] = None) -> Tensor:
    """y = W x (+bias)(+residual).  x [B,K,H,W]; w [M,K(,1,1)] or, transposed, [K,M(,1,1)] used as W^T.
    x2: optional second K-panel (channel concat without the concat).  f8 = (sx, sw): fp8 e4m3 MFMA operands (x / sx, w / sw,
    powers of two; bf16 tensors, wave-owned kernel forms only - the call fails elsewhere)."""
    _gpu(x, w, bias, residual, x2)
    _f32(w, "1x1 weight"); _f32(bias, "1x1 bias")
    B, K1, H, W = x.shape
    K2 = 0 if x2 is None else x2.shape[1]
    N = H * W
    w2 = w.reshape(w.shape[0], -1)
    M = w2.shape[1] if transposed else w2.shape[0]
    assert (w2.shape[0] if transposed else w2.shape[1]) == K1 + K2, "weight/input channel mismatch"
    y = torch.empty((B, M, H, W), dtype=x.dtype, device=x.device) if out is None else out
    assert y.shape == (B, M, H, W) and y.is_contiguous() and y.dtype == x.dtype
    d = L.PwDesc()
    d.x1, d.x1_bs, d.x1_gs, d.k1 = _p(x), K1 * N, 0, K1
    d.x2, d.x2_bs, d.x2_gs, d.k2 = _p(x2), K2 * N, 0, K2
    d.w, d.w_bs, d.w_gs = _p(w2), 0, 0
    d.w_sm, d.w_sk = (1, w2.shape[1]) if transposed else (w2.shape[1], 1)
    d.bias, d.bias_gs = _p(bias), 0
    d.r, d.r_bs, d.r_gs = _p(residual), M * N, 0
    d.y, d.y_bs, d.y_gs = _p(y), M * N, 0
    d.m, d.n, d.batch, d.groups, d.dtype = M, N, B, 1, _dt(x)
    if f8 is not None:
        d.f8, d.f8_sx, d.f8_sw = 1, float(f8[0]), float(f8[1])
    pw_gemm_desc(d, x.device)
    return y


def pw_gemm_desc(d: "L.PwDesc", device) -> None:
    """Run one mi_pw_gemm call described by ``d`` (allocates its weight-pack workspace)."""
    ws = _ws(L.lib().mi_pw_gemm_workspace(C.byref(d)), device)
    L.check(L.lib().mi_pw_gemm(C.byref(d), _p(ws), _stream()), "pw_gemm")


def gram(a: Tensor, b: Tensor, groups: int = 1, sum_batch: bool = False, want_sumsq: bool = False,
         out: Optional[Tensor] = None, accumulate: bool = False):
    """G[z][i][j] = sum_n a[z][i][n] b[z][j][n] with a,b [B, groups*m, H, W] split head-major into groups.
    out (+ accumulate): write / add the result into a caller's fp32 buffer of Z*ma*mb elements - a weight-gradient slot of the
    trainer's flat buffer; under the trainer's deferral window the final sum then joins the one table-driven reduction launch."""
    _gpu(a, b, out)
    B, Ca, H, W = a.shape
    Cb = b.shape[1]
    N = H * W
    ma, mb = Ca // groups, Cb // groups
    Z = groups if sum_batch else B * groups
    if out is None:
        if accumulate:
            raise ValueError("gram: accumulate needs the output buffer")
        out = torch.empty((Z, ma, mb), dtype=torch.float32, device=a.device)
    elif out.dtype != torch.float32 or out.numel() != Z * ma * mb:
        raise ValueError(f"gram: out must be float32 with {Z * ma * mb} elements")
    ss = torch.empty((B * groups, ma + mb), dtype=torch.float32, device=a.device) if want_sumsq else None
    d = L.GramDesc()
    d.a, d.a_bs, d.a_gs, d.ma = _p(a), Ca * N, ma * N, ma
    d.b, d.b_bs, d.b_gs, d.mb = _p(b), Cb * N, mb * N, mb
    d.n, d.batch, d.groups, d.dtype = N, B, groups, _dt(a)
    d.sum_batch, d.accumulate = int(sum_batch), int(accumulate)
    d.out, d.out_ld, d.out_zs, d.sumsq = _p(out), mb, ma * mb, _p(ss)
    ws = _ws(L.lib().mi_gram_workspace(C.byref(d)), a.device)
    L.check(L.lib().mi_gram(C.byref(d), _p(ws), _stream()), "gram")
    return (out, ss) if want_sumsq else out


# ----------------------------------------------------------------------------- MDTA / GDFN modules
MdtaParamsT = Tuple[Tensor, Tensor, Optional[Tensor], Tensor, Optional[Tensor], Tensor, Optional[Tensor]]
GdfnParamsT = Tuple[Tensor, Optional[Tensor], Tensor, Optional[Tensor], Tensor, Optional[Tensor]]


def _mdta_shape(x: Tensor, heads: int, ks: int) -> L.MdtaShape:
    B, Cc, H, W = x.shape
    return L.MdtaShape(B, Cc, heads, H, W, _dt(x), ks)


def _mdta_params(p: Sequence[Optional[Tensor]]) -> L.MdtaParams:
    for t in p:
        _f32(t, "MDTA parameter")
    return L.MdtaParams(*[_p(t) for t in p])


LnHeadT = Tuple[Tensor, Optional[Tensor], bool]     # LayerNorm weight, bias, want_stats


def _ln_head(ln: LnHeadT, x: Tensor):
    w, b, want_stats = ln
    _gpu(w, b)
    _f32(w, "LayerNorm weight"); _f32(b, "LayerNorm bias")
    mean = rstd = None
    if want_stats:
        mean = torch.empty((x.shape[0], x.shape[2] * x.shape[3]), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    return L.LnHead(_p(w), _p(b), _p(mean), _p(rstd), int(b is not None)), mean, rstd


def mdta_fwd_ln_ok(x: Tensor, heads: int, ks: int) -> bool:
    """Can norm1 run inside the qkv GEMM (mi_mdta_fwd_ln)?"""
    if not x.is_cuda:
        return False
    s = _mdta_shape(x, heads, ks)
    return bool(L.lib().mi_mdta_fwd_ln_ok(C.byref(s)))


F8ScalesT = Tuple[float, float, float, float]       # x1, w1 (first projection: input, weight), x2, w2 (second projection)


def mdta_fwd_f8_ok(x: Tensor, heads: int, ks: int, with_ln: bool) -> bool:
    """Can both MDTA projections run on fp8 MFMA operands (mi_mdta_fwd_f8)?"""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        return False
    s = _mdta_shape(x, heads, ks)
    return bool(L.lib().mi_mdta_fwd_f8_ok(C.byref(s), int(with_ln)))


def mdta_fwd(x: Tensor, residual: Optional[Tensor], params: MdtaParamsT, heads: int, need_saved: bool,
             ln: Optional[LnHeadT] = None, f8: Optional[F8ScalesT] = None):
    """params = (temperature, qkv.weight, qkv.bias, qkv_dwconv.weight, qkv_dwconv.bias, project_out.weight, .bias).
    ln = (weight, bias, want_stats): x is the LayerNorm INPUT and the norm runs inside the qkv GEMM; returns
    (out, saved, mean, rstd) then.  f8 = (x1, w1, x2, w2): inference with fp8 e4m3 MFMA operands in both projections
    (nothing saved, no statistics); returns out."""
    _gpu(x, residual, *params)
    ks = params[3].shape[-1]
    s = _mdta_shape(x, heads, ks)
    lib = L.lib()
    out = torch.empty_like(x)
    if f8 is not None:
        if need_saved or (ln is not None and ln[2]):
            raise ValueError("fp8 projections are an inference path: nothing can be saved")
        ws = _ws(lib.mi_mdta_workspace(C.byref(s)), x.device)
        lh = _ln_head(ln, x)[0] if ln is not None else None
        L.check(lib.mi_mdta_fwd_f8(C.byref(s), C.byref(_mdta_params(params)), C.byref(lh) if lh is not None else None,
                                   C.byref(L.F8Scales(*[float(v) for v in f8])), _p(x), _p(residual), _p(out), _p(ws),
                                   _stream()), "mdta_fwd_f8")
        return out
    saved = _blob(lib.mi_mdta_saved_bytes(C.byref(s)), x.device) if need_saved else None
    ws = _ws(lib.mi_mdta_workspace(C.byref(s)), x.device)
    pp = _mdta_params(params)
    if ln is None:
        L.check(lib.mi_mdta_fwd(C.byref(s), C.byref(pp), _p(x), _p(residual), _p(out), _p(saved), _p(ws), _stream()),
                "mdta_fwd")
        return out, saved
    lh, mean, rstd = _ln_head(ln, x)
    L.check(lib.mi_mdta_fwd_ln(C.byref(s), C.byref(pp), C.byref(lh), _p(x), _p(residual), _p(out), _p(saved), _p(ws),
                               _stream()), "mdta_fwd_ln")
    return out, saved, mean, rstd


LnTailT = Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor], Tensor, Tensor]   # w, b, mean, rstd, dres, dw, db


def _ln_tail(ln: LnTailT, like: Tensor) -> L.LnTail:
    w, b, mean, rstd, dres, dw, db = ln
    _gpu(w, b, mean, rstd, dres, dw, db)
    for t_ in (w, b, mean, rstd, dw, db):
        _f32(t_, "LayerNorm tail tensor")
    if dres is not None and (dres.shape != like.shape or dres.dtype != like.dtype):
        raise ValueError("LayerNorm tail: dres must match the block input")
    return L.LnTail(_p(w), _p(b), _p(mean), _p(rstd), _p(dres), _p(dw), _p(db))


def mdta_bwd_ln_ok(x: Tensor, heads: int, ks: int, qkv_bias: bool) -> bool:
    """Can the MDTA backward end in the one-launch tail (qkv weight gradient + W^T dY + LayerNorm backward + residual)?"""
    if not x.is_cuda:
        return False
    s = _mdta_shape(x, heads, ks)
    return bool(L.lib().mi_mdta_bwd_ln_ok(C.byref(s), int(qkv_bias)))


def mdta_bwd(x: Tensor, dout: Tensor, params: MdtaParamsT, heads: int, saved: Tensor,
             grads: Sequence[Optional[Tensor]], accumulate: bool, ln: Optional[LnTailT] = None) -> Tensor:
    """ln = None: x is the LayerNorm output (the conv input), the result its gradient.  ln given: x is the LayerNorm INPUT
    and the result the gradient of the half-block's input (mi_mdta_bwd_ln)."""
    _gpu(x, dout, saved, *params, *grads)
    ks = params[3].shape[-1]
    s = _mdta_shape(x, heads, ks)
    lib = L.lib()
    dx = torch.empty_like(x)
    pp = _mdta_params(params)
    for t in grads:
        _f32(t, "MDTA gradient")
    gg = L.MdtaGrads(*[_p(t) for t in grads], int(accumulate))
    if ln is None:
        ws = _ws(lib.mi_mdta_workspace(C.byref(s)), x.device)
        L.check(lib.mi_mdta_bwd(C.byref(s), C.byref(pp), _p(x), _p(dout), _p(dx), C.byref(gg), _p(saved), _p(ws), _stream()),
                "mdta_bwd")
    else:
        lt = _ln_tail(ln, x)
        ws = _ws(lib.mi_mdta_bwd_ln_workspace(C.byref(s)), x.device)
        L.check(lib.mi_mdta_bwd_ln(C.byref(s), C.byref(pp), C.byref(lt), _p(x), _p(dout), _p(dx), C.byref(gg), _p(saved),
                                   _p(ws), _stream()), "mdta_bwd_ln")
    return dx


def _xmdta_shape(x: Tensor, heads: int, ks_q: int, ks_kv: int) -> L.XmdtaShape:
    B, Cc, H, W = x.shape
    return L.XmdtaShape(B, Cc, heads, H, W, _dt(x), ks_q, ks_kv)


def xmdta_fwd(x: Tensor, y: Tensor, residual: Optional[Tensor], params: Sequence[Optional[Tensor]], heads: int,
              need_saved: bool):
    """Cross-MDTA.  params = (temperature, q.weight, q.bias, q_dwconv.weight, q_dwconv.bias, kv.weight, kv.bias,
    kv_dwconv.weight, kv_dwconv.bias, project_out.weight, project_out.bias)."""
    _gpu(x, y, residual, *params)
    assert x.shape == y.shape and x.dtype == y.dtype, "cross-MDTA needs x and y of one shape and dtype"
    for t in params:
        _f32(t, "cross-MDTA parameter")
    s = _xmdta_shape(x, heads, params[3].shape[-1], params[7].shape[-1])
    lib = L.lib()
    out = torch.empty_like(x)
    saved = _blob(lib.mi_xmdta_saved_bytes(C.byref(s)), x.device) if need_saved else None
    ws = _ws(lib.mi_xmdta_workspace(C.byref(s)), x.device)
    pp = L.XmdtaParams(*[_p(t) for t in params])
    L.check(lib.mi_xmdta_fwd(C.byref(s), C.byref(pp), _p(x), _p(y), _p(residual), _p(out), _p(saved), _p(ws), _stream()),
            "xmdta_fwd")
    return out, saved


def xmdta_bwd(x: Tensor, y: Tensor, dout: Tensor, params: Sequence[Optional[Tensor]], heads: int, saved: Tensor,
              grads: Sequence[Optional[Tensor]], accumulate: bool):
    _gpu(x, y, dout, saved, *params, *grads)
    s = _xmdta_shape(x, heads, params[3].shape[-1], params[7].shape[-1])
    lib = L.lib()
    dx, dy = torch.empty_like(x), torch.empty_like(y)
    ws = _ws(lib.mi_xmdta_workspace(C.byref(s)), x.device)
    pp = L.XmdtaParams(*[_p(t) for t in params])
    for t in grads:
        _f32(t, "cross-MDTA gradient")
    gg = L.XmdtaGrads(*[_p(t) for t in grads], int(accumulate))
    L.check(lib.mi_xmdta_bwd(C.byref(s), C.byref(pp), _p(x), _p(y), _p(dout), _p(dx), _p(dy), C.byref(gg), _p(saved), _p(ws),
                             _stream()), "xmdta_bwd")
    return dx, dy


def _gdfn_shape(x: Tensor, hidden: int, ks: int, flags: int = 0) -> L.GdfnShape:
    B, Cc, H, W = x.shape
    return L.GdfnShape(B, Cc, hidden, H, W, _dt(x), ks, flags)


def gdfn_fwd_ln_ok(x: Tensor, hidden: int, ks: int) -> bool:
    if not x.is_cuda:
        return False
    s = _gdfn_shape(x, hidden, ks, 0)
    return bool(L.lib().mi_gdfn_fwd_ln_ok(C.byref(s)))


def gdfn_fwd_f8_ok(x: Tensor, hidden: int, ks: int, with_ln: bool) -> bool:
    if not x.is_cuda or x.dtype != torch.bfloat16:
        return False
    s = _gdfn_shape(x, hidden, ks, 0)
    return bool(L.lib().mi_gdfn_fwd_f8_ok(C.byref(s), int(with_ln)))


def gdfn_fwd(x: Tensor, residual: Optional[Tensor], params: GdfnParamsT, need_saved: bool, ln: Optional[LnHeadT] = None,
             f8: Optional[F8ScalesT] = None):
    """params = (project_in.weight, .bias, dwconv.weight, .bias, project_out.weight, .bias).  ln, f8: as in mdta_fwd."""
    _gpu(x, residual, *params)
    for t in params:
        _f32(t, "GDFN parameter")
    hidden, ks = params[4].shape[1], params[2].shape[-1]
    if f8 is not None:
        if need_saved or (ln is not None and ln[2]):
            raise ValueError("fp8 projections are an inference path: nothing can be saved")
        s = _gdfn_shape(x, hidden, ks, 0)
        lib = L.lib()
        out = torch.empty_like(x)
        ws = _ws(lib.mi_gdfn_workspace(C.byref(s)), x.device)
        lh = _ln_head(ln, x)[0] if ln is not None else None
        L.check(lib.mi_gdfn_fwd_f8(C.byref(s), C.byref(L.GdfnParams(*[_p(t) for t in params])),
                                   C.byref(lh) if lh is not None else None, C.byref(L.F8Scales(*[float(v) for v in f8])),
                                   _p(x), _p(residual), _p(out), _p(ws), _stream()), "gdfn_fwd_f8")
        return out
    # A/B switch: keep the conv output instead of recomputing it in backward.  Read here, once per forward; backward
    # recovers the choice from the blob's size, so toggling the variable between the two cannot desynchronise them.
    s = _gdfn_shape(x, hidden, ks, 1 if env("MI_GDFN_STORE_Y") else 0)
    lib = L.lib()
    out = torch.empty_like(x)
    saved = _blob(lib.mi_gdfn_saved_bytes(C.byref(s)), x.device) if need_saved else None
    ws = _ws(lib.mi_gdfn_workspace(C.byref(s)), x.device)
    pp = L.GdfnParams(*[_p(t) for t in params])
    if ln is None:
        L.check(lib.mi_gdfn_fwd(C.byref(s), C.byref(pp), _p(x), _p(residual), _p(out), _p(saved), _p(ws), _stream()),
                "gdfn_fwd")
        return out, saved
    lh, mean, rstd = _ln_head(ln, x)
    L.check(lib.mi_gdfn_fwd_ln(C.byref(s), C.byref(pp), C.byref(lh), _p(x), _p(residual), _p(out), _p(saved), _p(ws),
                               _stream()), "gdfn_fwd_ln")
    return out, saved, mean, rstd


def gdfn_bwd_ln_ok(x: Tensor, hidden: int, ks: int, in_bias: bool) -> bool:
    if not x.is_cuda:
        return False
    s = _gdfn_shape(x, hidden, ks, 0)
    return bool(L.lib().mi_gdfn_bwd_ln_ok(C.byref(s), int(in_bias)))


def gdfn_bwd(x: Tensor, dout: Tensor, params: GdfnParamsT, saved: Tensor, grads: Sequence[Optional[Tensor]],
             accumulate: bool, ln: Optional[LnTailT] = None) -> Tensor:
    """ln: as in mdta_bwd (x is then the LayerNorm INPUT; mi_gdfn_bwd_ln)."""
    _gpu(x, dout, saved, *params, *grads)
    hidden, ks = params[4].shape[1], params[2].shape[-1]
    lib = L.lib()
    s = None
    for flags in (0, 1):                   # which layout did the forward carve?  the two differ in size
        cand = _gdfn_shape(x, hidden, ks, flags)
        if max(int(lib.mi_gdfn_saved_bytes(C.byref(cand))), 256) == saved.numel():
            s = cand
            break
    if s is None:
        raise RuntimeError("gdfn_bwd: the saved blob matches neither layout of this shape")
    dx = torch.empty_like(x)
    pp = L.GdfnParams(*[_p(t) for t in params])
    for t in grads:
        _f32(t, "GDFN gradient")
    gg = L.GdfnGrads(*[_p(t) for t in grads], int(accumulate))
    if ln is None:
        ws = _ws(lib.mi_gdfn_workspace(C.byref(s)), x.device)
        L.check(lib.mi_gdfn_bwd(C.byref(s), C.byref(pp), _p(x), _p(dout), _p(dx), C.byref(gg), _p(saved), _p(ws), _stream()),
                "gdfn_bwd")
    else:
        lt = _ln_tail(ln, x)
        ws = _ws(lib.mi_gdfn_bwd_ln_workspace(C.byref(s)), x.device)
        L.check(lib.mi_gdfn_bwd_ln(C.byref(s), C.byref(pp), C.byref(lt), _p(x), _p(dout), _p(dx), C.byref(gg), _p(saved),
                                   _p(ws), _stream()), "gdfn_bwd_ln")
    return dx


def bwd_tail_ok(M: int, Cc: int, N: int, dtype: torch.dtype) -> bool:
    return bool(L.lib().mi_bwd_tail_ok(M, Cc, N, L.MI_BF16 if dtype == torch.bfloat16 else L.MI_F32))


def bwd_tail(dy: Tensor, x: Tensor, dres: Optional[Tensor], mean: Tensor, rstd: Tensor, w: Tensor, gamma: Tensor,
             beta: Tensor, dw: Tensor, dgamma: Tensor, dbeta: Tensor, accumulate: bool) -> Tensor:
    """The backward tail by itself (mi_bwd_tail): dy [B,M,H,W], x the LayerNorm input [B,C,H,W]; returns dx."""
    _gpu(dy, x, dres, mean, rstd, w, gamma, beta, dw, dgamma, dbeta)
    B, Cc, H, W = x.shape
    M = dy.shape[1]
    dx = torch.empty_like(x)
    lib = L.lib()
    ws = _ws(lib.mi_bwd_tail_workspace(M, Cc), x.device)
    L.check(lib.mi_bwd_tail(_p(dy), M, _p(x), Cc, _p(dres), _p(mean), _p(rstd), _p(w), _p(gamma), _p(beta), _p(dx), _p(dw),
                            _p(dgamma), _p(dbeta), B, H * W, int(accumulate), _dt(x), _p(ws), _stream()), "bwd_tail")
    return dx


# ----------------------------------------------------------------------------- fused half-blocks (bf16)
def _gdfn_fused_shape(x: Tensor, hidden: int, with_bias: bool) -> L.GdfnFusedShape:
    B, Cc, H, W = x.shape
    return L.GdfnFusedShape(B, Cc, hidden, H, W, int(with_bias))


def gdfn_fused_ok(x: Tensor, hidden: int, ks: int = 3) -> bool:
    """True when the one-launch LN + GDFN + residual kernel covers this activation (bf16, 3x3, tile-aligned)."""
    if x.dtype != torch.bfloat16 or ks != 3 or not x.is_cuda:
        return False
    s = _gdfn_fused_shape(x, hidden, True)
    return bool(L.lib().mi_gdfn_fused_ok(C.byref(s)))


def gdfn_fused_pack(x_like: Tensor, ln_w: Tensor, ln_b: Optional[Tensor], params: "GdfnParamsT") -> Tensor:
    """LayerNorm affine + GDFN parameters -> the fused kernel's packed weight images (re-run after weight updates)."""
    _gpu(ln_w, ln_b, *params)
    for t in (ln_w, ln_b) + tuple(params):
        _f32(t, "fused GDFN parameter")
    hidden = params[4].shape[1]
    s = _gdfn_fused_shape(x_like, hidden, ln_b is not None)
    lib = L.lib()
    pack = _blob(lib.mi_gdfn_fused_pack_bytes(C.byref(s)), ln_w.device)
    pp = L.GdfnParams(*[_p(t) for t in params])
    L.check(lib.mi_gdfn_fused_pack(C.byref(s), _p(ln_w), _p(ln_b), C.byref(pp), _p(pack), _stream()), "gdfn_fused_pack")
    return pack


def gdfn_fused_fwd(y: Tensor, pack: Tensor, hidden: int, with_bias: bool, want_stats: bool = False,
                   f8: Optional[F8ScalesT] = None):
    """out = y + GDFN(LN(y)) in one launch; optionally the LN statistics [B, H*W] of y.  f8 = (x1, w1, x2, w2): fp8 e4m3 MFMA
    operands in both projections (inference: no statistics); x1 scales the normalised input, w1 the packed W_in . diag(gamma)."""
    _gpu(y, pack)
    s = _gdfn_fused_shape(y, hidden, with_bias)
    out = torch.empty_like(y)
    if f8 is not None:
        if want_stats:
            raise ValueError("fp8 projections are an inference path: no statistics")
        L.check(L.lib().mi_gdfn_fused_fwd_f8(C.byref(s), _p(pack), C.byref(L.F8Scales(*[float(v) for v in f8])), _p(y), _p(out),
                                             _stream()), "gdfn_fused_fwd_f8")
        return out, None, None
    mean = rstd = None
    if want_stats:
        mean = torch.empty((y.shape[0], y.shape[2] * y.shape[3]), dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
    L.check(L.lib().mi_gdfn_fused_fwd(C.byref(s), _p(pack), _p(y), _p(out), _p(mean), _p(rstd), _stream()), "gdfn_fused_fwd")
    return out, mean, rstd


def gdfn_fused_train_ok(x: Tensor, hidden: int, ks: int = 3) -> bool:
    """True when the one-launch LN + GDFN forward can also write what the backward reads (mi_gdfn_fused_fwd_train)."""
    if x.dtype != torch.bfloat16 or ks != 3 or not x.is_cuda:
        return False
    return bool(L.lib().mi_gdfn_fused_fwd_train_ok(C.byref(_gdfn_fused_shape(x, hidden, True))))


def gdfn_fused_fwd_train(y: Tensor, pack: Tensor, hidden: int, with_bias: bool):
    """out = y + GDFN(LN(y)) in one launch on the TRAINING path -> (out, saved, mean, rstd): ``saved`` is the blob gdfn_bwd
    reads (project_in output + gate output), mean / rstd the LayerNorm statistics of y."""
    _gpu(y, pack)
    s = _gdfn_fused_shape(y, hidden, with_bias)
    lib = L.lib()
    out = torch.empty_like(y)
    mean = torch.empty((y.shape[0], y.shape[2] * y.shape[3]), dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    saved = _blob(lib.mi_gdfn_saved_bytes(C.byref(_gdfn_shape(y, hidden, 3, 0))), y.device)
    L.check(lib.mi_gdfn_fused_fwd_train(C.byref(s), _p(pack), _p(y), _p(out), _p(mean), _p(rstd), _p(saved), _stream()),
            "gdfn_fused_fwd_train")
    return out, saved, mean, rstd


def mdta_fused_ok(x: Tensor, heads: int, ks: int = 3) -> bool:
    """True when the one-launch LN -> qkv -> dw3x3 -> q k^T pass (csrc/fused_mdta.hip) covers this activation."""
    if x.dtype != torch.bfloat16 or ks != 3 or not x.is_cuda:
        return False
    return bool(L.lib().mi_mdta_fused_ok(C.byref(_mdta_shape(x, heads, ks))))


def mdta_fused_pays(x: Tensor, heads: int, ks: int = 3) -> bool:
    """Covered and large enough (one workgroup for nearly every CU) for the fused pass to beat the unfused chain."""
    if x.dtype != torch.bfloat16 or ks != 3 or not x.is_cuda:
        return False
    return bool(L.lib().mi_mdta_fused_pays(C.byref(_mdta_shape(x, heads, ks))))


def mdta_fused_pack(x_like: Tensor, heads: int, ln_w: Tensor, ln_b: Optional[Tensor], params: "MdtaParamsT") -> Tensor:
    """LayerNorm affine + qkv / depthwise parameters -> the fused MDTA kernel's packed weight images."""
    _gpu(ln_w, ln_b, *params)
    for t in (ln_w, ln_b) + tuple(params):
        _f32(t, "fused MDTA parameter")
    s = _mdta_shape(x_like, heads, params[3].shape[-1])
    lib = L.lib()
    pack = _blob(lib.mi_mdta_fused_pack_bytes(C.byref(s)), ln_w.device)
    L.check(lib.mi_mdta_fused_pack(C.byref(s), _p(ln_w), _p(ln_b), C.byref(_mdta_params(params)), _p(pack), _stream()),
            "mdta_fused_pack")
    return pack


def mdta_fused_fwd(x: Tensor, pack: Tensor, params: "MdtaParamsT", heads: int, with_bias: bool, residual: Optional[Tensor],
                   want_stats: bool = False):
    """out = residual + MDTA(LN(x)) with the whole producer chain of q k^T in one launch (nothing saved: the no_grad path).
    -> (out, mean, rstd)."""
    _gpu(x, pack, residual, *params)
    s = _mdta_shape(x, heads, params[3].shape[-1])
    lib = L.lib()
    out = torch.empty_like(x)
    mean = rstd = None
    if want_stats:
        mean = torch.empty((x.shape[0], x.shape[2] * x.shape[3]), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    ws = _ws(lib.mi_mdta_fused_workspace(C.byref(s)), x.device)
    L.check(lib.mi_mdta_fused_fwd(C.byref(s), C.byref(_mdta_params(params)), _p(pack), int(with_bias), _p(x), _p(residual),
                                  _p(out), _p(mean), _p(rstd), _p(ws), _stream()), "mdta_fused_fwd")
    return out, mean, rstd


# ----------------------------------------------------------------------------- router GAP
def rows_gather(x: Tensor, idx: Tensor) -> Tensor:
    """out[i] = x[idx[i]] over whole [C,H,W] rows (SparseDispatcher.dispatch)."""
    _gpu(x, idx)
    assert idx.dtype == torch.int64 and x.is_contiguous()
    n = int(idx.numel())
    out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    row = x[0].numel()
    L.check(L.lib().mi_rows_gather(_p(x), _p(idx), _p(out), n, row, _dt(x), _stream()), "rows_gather")
    return out


def rows_gather_scaled(x_f32: Tensor, idx: Tensor, scale: Optional[Tensor], dtype: torch.dtype) -> Tensor:
    """out[i] = scale[i] * x_f32[idx[i]] cast to `dtype`."""
    _gpu(x_f32, idx, scale)
    assert x_f32.dtype == torch.float32 and idx.dtype == torch.int64 and x_f32.is_contiguous()
    n = int(idx.numel())
    out = torch.empty((n,) + tuple(x_f32.shape[1:]), dtype=dtype, device=x_f32.device)
    L.check(L.lib().mi_rows_gather_scaled(_p(x_f32), _p(idx), _p(_f32(scale, "scale")), _p(out), n, x_f32[0].numel(),
                                          L.MI_BF16 if dtype == torch.bfloat16 else L.MI_F32, _stream()), "rows_gather_scaled")
    return out


def rows_scatter_add(src: Tensor, idx: Tensor, scale: Optional[Tensor], n_rows: int, out_f32: bool) -> Tensor:
    """out[b] = sum_{i: idx[i]==b} scale[i] * src[i], accumulated in fp32 (SparseDispatcher.combine)."""
    _gpu(src, idx, scale)
    assert idx.dtype == torch.int64 and src.is_contiguous()
    out = torch.empty((n_rows,) + tuple(src.shape[1:]), dtype=torch.float32 if out_f32 else src.dtype, device=src.device)
    row = src[0].numel() if src.shape[0] else out[0].numel()
    L.check(L.lib().mi_rows_scatter_add(_p(src), _p(idx), _p(_f32(scale, "scale")), _p(out), int(src.shape[0]), n_rows, row,
                                        _dt(src), 1 if out_f32 else 0, _stream()), "rows_scatter_add")
    return out


def rows_dot(g: Tensor, src: Tensor, idx: Tensor) -> Tensor:
    """out[i] = <g[idx[i]], src[i]> (fp32 g): gradient of the combine's gate values."""
    _gpu(g, src, idx)
    n = int(src.shape[0])
    out = torch.zeros(n, dtype=torch.float32, device=src.device)
    if n == 0:
        return out
    row = src[0].numel()
    ws = _ws(L.lib().mi_rows_dot_workspace(n, row), src.device)
    L.check(L.lib().mi_rows_dot(_p(_f32(g, "g")), _p(src), _p(idx), _p(out), n, row, _dt(src), _p(ws), _stream()), "rows_dot")
    return out


def glue3x3_ok(H: int, W: int) -> bool:
    return bool(L.lib().mi_glue3x3_ok(H, W))


def im2col3x3(x: Tensor, flip: bool = False) -> Tensor:
    """x[B,C,H,W] -> [B,9C,H,W] with col[c*9+ky*3+kx][y][x] = x[c][y+ky-1][x+kx-1] (shifts negated when flip)."""
    _gpu(x)
    B, Cn, H, W = x.shape
    out = torch.empty((B, 9 * Cn, H, W), dtype=x.dtype, device=x.device)
    L.check(L.lib().mi_im2col3x3(_p(x), _p(out), B, Cn, H, W, 1 if flip else 0, _dt(x), _stream()), "im2col3x3")
    return out


def col2im3x3(z: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, flip: bool = False) -> Tensor:
    """z[B,9M,H,W] -> y[B,M,H,W], y[m] = sum over taps of the tap plane shifted back (+ bias[m]) (+ residual);
    flip negates the shifts (transposed convolution)."""
    _gpu(z, bias, residual)
    B, M9, H, W = z.shape
    M = M9 // 9
    y = torch.empty((B, M, H, W), dtype=z.dtype, device=z.device)
    L.check(L.lib().mi_col2im3x3(_p(z), _p(_f32(bias, "bias")), _p(residual), _p(y), B, M, H, W, 1 if flip else 0, _dt(z), _stream()), "col2im3x3")
    return y


def chan_sum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """Bias gradient of a conv: out[c] (+)= sum over batch and pixels of x[b, c] (fp32, fixed order)."""
    _gpu(x, out)
    B, Cn, H, W = x.shape
    if out is None:
        if accumulate:
            raise ValueError("chan_sum: accumulate needs the gradient buffer")
        out = torch.empty(Cn, dtype=torch.float32, device=x.device)
    _f32(out, "bias gradient")
    lib = L.lib()
    ws = _ws(lib.mi_chan_sum_workspace(Cn, H * W), x.device)
    L.check(lib.mi_chan_sum(_p(x), _p(out), B, Cn, H * W, _dt(x), 1 if accumulate else 0, _p(ws), _stream()), "chan_sum")
    return out


def conv3x3_ok(x: Tensor) -> bool:
    """The implicit-GEMM 3x3 convolution covers this activation (bf16, W % 8 == 0; csrc/conv3x3.hip)."""
    return bool(x.is_cuda and x.dim() == 4 and L.lib().mi_conv3x3_ok(x.shape[2], x.shape[3], _dt(x)))


def conv3x3(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
            transpose: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Dense 3x3 conv (stride 1, pad 1) of x [B,K,H,W] as an implicit GEMM.  transpose False: weight [M,K,3,3], y = conv(x).
    transpose True: weight is the conv's own [K,M,3,3] and the result is its data gradient for the upstream gradient x.
    x / residual / out may be channel slices (dense [C,H,W] blocks, any batch stride)."""
    _gpu(weight, bias)
    for tsr in (x, residual, out):                       # channel slices allowed: checked for dense [C,H,W] blocks below
        if tsr is not None and not tsr.is_cuda:
            raise RuntimeError("image_restoration_amd ops run on the MI355X only (got a CPU tensor)")
    _f32(weight, "conv weight")
    B, K, H, W = x.shape
    M = weight.shape[1] if transpose else weight.shape[0]
    if (weight.shape[0] if transpose else weight.shape[1]) != K or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError(f"conv3x3: weight {tuple(weight.shape)} does not match {K} input channels")
    for tsr, name in ((x, "x"), (residual, "residual"), (out, "out")):
        if tsr is not None and not (tsr.stride(3) == 1 and tsr.stride(2) == W and tsr.stride(1) == H * W):
            raise ValueError(f"conv3x3: {name} must have dense [C,H,W] blocks")
    lib = L.lib()
    w = weight.contiguous()
    pack = _blob(lib.mi_conv3x3_pack_bytes(M, K), x.device)
    L.check(lib.mi_conv3x3_pack(_p(w), M, K, 1 if transpose else 0, _p(pack), _stream()), "conv3x3_pack")
    y = out if out is not None else torch.empty((B, M, H, W), dtype=x.dtype, device=x.device)
    L.check(lib.mi_conv3x3_fwd(_p(pack), _p(x), x.stride(0), _p(_f32(bias, "bias")), _p(residual),
                               residual.stride(0) if residual is not None else 0, _p(y), y.stride(0), B, M, K, H, W, _stream()),
            "conv3x3_fwd")
    return y


def conv3x3_wgrad(dy: Tensor, x: Tensor, dw: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """Weight gradient of the dense 3x3 conv: dw[M,K,3,3] (+)= sum over batch and pixels of dy[b,m,p] x[b,k,p+d(tap)] without the
    im2col expansion.  dy [B,M,H,W], x [B,K,H,W] bf16 (channel slices allowed); dw fp32 (allocated when None)."""
    _gpu(dw)
    for tsr, name in ((dy, "dy"), (x, "x")):
        if not tsr.is_cuda:
            raise RuntimeError("image_restoration_amd ops run on the MI355X only (got a CPU tensor)")
        if not (tsr.stride(3) == 1 and tsr.stride(2) == tsr.shape[3] and tsr.stride(1) == tsr.shape[2] * tsr.shape[3]):
            raise ValueError(f"conv3x3_wgrad: {name} must have dense [C,H,W] blocks")
    B, M, H, W = dy.shape
    K = x.shape[1]
    if dw is None:
        if accumulate:
            raise ValueError("conv3x3_wgrad: accumulate needs the gradient buffer")
        dw = torch.empty((M, K, 3, 3), dtype=torch.float32, device=dy.device)
    _f32(dw, "conv weight gradient")
    lib = L.lib()
    ws = _ws(lib.mi_conv3x3_wgrad_workspace(B, M, K, H, W), dy.device)
    L.check(lib.mi_conv3x3_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), 1 if accumulate else 0, B, M, K, H, W, _p(ws),
                                 _stream()), "conv3x3_wgrad")
    return dw


def pixel_shuffle2(x: Tensor, unshuffle: bool, out: Optional[Tensor] = None) -> Tensor:
    """PixelShuffle(2) (unshuffle False: [B,4c,H,W] -> [B,c,2H,2W]) or its inverse.  x / out may be channel slices of wider
    tensors (dense [C,H,W] blocks, any batch stride)."""
    if not x.is_cuda:
        raise RuntimeError("image_restoration_amd ops run on the MI355X only (got a CPU tensor)")
    B, Cx, Hx, Wx = x.shape
    if unshuffle:
        c, H, W = Cx, Hx // 2, Wx // 2
        shape = (B, 4 * c, H, W)
    else:
        c, H, W = Cx // 4, Hx, Wx
        shape = (B, c, 2 * H, 2 * W)
    if out is None:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    assert tuple(out.shape) == shape and out.dtype == x.dtype
    L.check(L.lib().mi_pixel_shuffle2(_p(x), _bstride(x), _p(out), _bstride(out), B, c, H, W, int(unshuffle), _dt(x), _stream()),
            "pixel_shuffle2")
    return out


def copy_rows(src: Tensor, dst: Tensor) -> Tensor:
    """dst <- src for [B,C,H,W] tensors whose [C,H,W] blocks are dense (either may be a channel slice)."""
    B = src.shape[0]
    L.check(L.lib().mi_copy_rows(_p(src), _bstride(src), _p(dst), _bstride(dst), B, src[0].numel(), _dt(src), _stream()), "copy_rows")
    return dst


def gap_fwd(x: Tensor) -> Tensor:
    _gpu(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    L.check(L.lib().mi_gap_fwd(_p(x), _p(out), B, Cc, H * W, _dt(x), _stream()), "gap_fwd")
    return out


def gap_bwd(dout: Tensor, like: Tensor) -> Tensor:
    dout = dout.contiguous()
    _gpu(dout, like)
    B, Cc, H, W = like.shape
    dx = torch.empty_like(like)
    L.check(L.lib().mi_gap_bwd(_p(dout.contiguous().float()), _p(dx), B, Cc, H * W, _dt(like), _stream()), "gap_bwd")
    return dx


# ----------------------------------------------------------------------------- MoCE kernels (csrc/moce.hip)
class RouteTables:
    """Device-side SparseDispatcher bookkeeping produced by the router launch (no nonzero / sort / tolist)."""
    __slots__ = ("counts", "offsets", "perm", "perm_gate", "perm_expert", "row_of", "logits")


def moe_route_fwd(pooled: Tensor, freq: Tensor, wg: Tensor, wf: Tensor, noise: Tensor, complexity: Optional[Tensor], k: int,
                  training: bool):
    """-> gates [B,E], topk_idx [B,k] int64, topk_val [B,k], aux [1], RouteTables."""
    _gpu(pooled, freq, wg, wf, noise, complexity)
    for t in (pooled, freq, wg, wf, noise, complexity):
        _f32(t, "router tensor")
    B, Cc = pooled.shape
    Fd, E = freq.shape[1], wg.shape[0]
    dev = pooled.device
    f32 = dict(dtype=torch.float32, device=dev)
    gates, aux = torch.empty((B, E), **f32), torch.empty(1, **f32)
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    val = torch.empty((B, k), **f32)
    tb = RouteTables()
    tb.logits = torch.empty((B, E), **f32)
    tb.counts = torch.empty(E, dtype=torch.int32, device=dev)
    tb.offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    tb.perm = torch.empty(B * k, dtype=torch.int64, device=dev)
    tb.perm_gate = torch.empty(B * k, **f32)
    tb.perm_expert = torch.empty(B * k, dtype=torch.int32, device=dev)
    tb.row_of = torch.empty((B, k), dtype=torch.int32, device=dev)
    L.check(L.lib().mi_moe_route_fwd(_p(pooled), _p(freq), _p(wg), _p(wf), _p(noise), _p(complexity), _p(tb.logits), _p(gates),
                                     _p(idx), _p(val), _p(aux), _p(tb.counts), _p(tb.offsets), _p(tb.perm), _p(tb.perm_gate),
                                     _p(tb.perm_expert), _p(tb.row_of), B, Cc, Fd, E, k, int(training), _stream()),
            "moe_route_fwd")
    return gates, idx, val, aux, tb


def moe_route_bwd(pooled: Tensor, freq: Tensor, wg: Tensor, wf: Tensor, noise: Tensor, complexity: Optional[Tensor],
                  tb: RouteTables, idx: Tensor, dgates: Optional[Tensor], drow: Optional[Tensor], daux: Optional[Tensor],
                  training: bool):
    _gpu(pooled, freq, wg, wf, noise, complexity, dgates, drow, daux)
    B, Cc = pooled.shape
    Fd, E, k = freq.shape[1], wg.shape[0], idx.shape[1]
    dpooled, dfreq, dwg, dwf = (torch.empty_like(t) for t in (pooled, freq, wg, wf))
    L.check(L.lib().mi_moe_route_bwd(_p(pooled), _p(freq), _p(wg), _p(wf), _p(noise), _p(complexity), _p(tb.logits), _p(idx),
                                     _p(_f32(dgates, "dgates")), _p(_f32(drow, "drow")), _p(tb.row_of), _p(_f32(daux, "daux")),
                                     _p(dpooled), _p(dfreq), _p(dwg), _p(dwf), B, Cc, Fd, E, k, int(training), _stream()),
            "moe_route_bwd")
    return dpooled, dfreq, dwg, dwf


def grouped_pw_gemm(problems, counts: Tensor, offsets: Tensor, max_rows: int, n_pix: int, dtype: torch.dtype) -> None:
    """ONE launch for the 1x1 projections of all experts (csrc/grouped.hip).  ``problems``: list of dicts with keys
    x, w (2-D fp32 [M, K] or, with transposed=True, [K, M] used as its transpose), y, m, k, expert and optionally bias, r,
    x_local / y_local / r_local (the buffer is indexed by the row's position inside its expert's segment instead of the
    stitched row number).  counts / offsets: the router's int32 DEVICE tables - no segment size is read by the host here."""
    _gpu(counts, offsets)
    assert counts.dtype == torch.int32 and offsets.dtype == torch.int32
    arr = (L.GroupedProblem * len(problems))()
    for i, q in enumerate(problems):
        w = q["w"]
        _f32(w, "expert weight")
        g = arr[i]
        g.x, g.x_rs = q["x"].data_ptr(), int(q.get("x_rs", 0))
        g.w = w.data_ptr()
        g.w_sm, g.w_sk = (1, w.stride(0)) if q.get("transposed") else (w.stride(0), 1)
        g.bias = _p(q.get("bias"))
        g.r, g.r_rs = _p(q.get("r")), int(q.get("r_rs", 0))
        g.y, g.y_rs = q["y"].data_ptr(), int(q.get("y_rs", 0))
        g.m, g.k, g.expert = int(q["m"]), int(q["k"]), int(q["expert"])
        g.x_local, g.y_local, g.r_local = int(bool(q.get("x_local"))), int(bool(q.get("y_local"))), int(bool(q.get("r_local")))
    L.check(L.lib().mi_grouped_pw_gemm(arr, len(problems), _p(counts), _p(offsets), int(max_rows), int(n_pix),
                                       L.MI_BF16 if dtype == torch.bfloat16 else L.MI_F32, _stream()), "grouped_pw_gemm")


def _bstride(t: Tensor) -> int:
    """Batch stride (elements) of a [B,C,H,W] tensor whose [C,H,W] block is dense (a channel slice of a wider tensor)."""
    B, Cc, H, W = t.shape
    assert t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W, "need dense [C,H,W] blocks"
    return t.stride(0) if B > 1 else Cc * H * W


def patch_circconv(x: Tensor, y: Tensor, patch: int, flip: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Per patch x patch block: irfft2(rfft2(x) * rfft2(y)) = 2-D circular convolution (zero padded to the patch grid).
    x, y: [B,C,H,W] (channel slices allowed); flip convolves with the index-reversed y (gradient form)."""
    for t in (x, y):
        if not t.is_cuda:
            raise RuntimeError("image_restoration_amd ops run on the MI355X only (got a CPU tensor)")
    B, Cc, H, W = x.shape
    assert x.shape == y.shape and x.dtype == y.dtype
    if out is None:
        out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    L.check(L.lib().mi_patch_circconv(_p(x), _bstride(x), _p(y), _bstride(y), _p(out), _bstride(out), B, Cc, H, W, int(patch),
                                      int(flip), _dt(x), _stream()), "patch_circconv")
    return out


def gelu_gap_fwd(x: Tensor) -> Tensor:
    _gpu(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    L.check(L.lib().mi_gelu_gap_fwd(_p(x), _p(out), B, Cc, H * W, _dt(x), _stream()), "gelu_gap_fwd")
    return out


def gelu_gap_bwd(x: Tensor, dout: Tensor) -> Tensor:
    dout = dout.contiguous()
    _gpu(x, dout)
    B, Cc, H, W = x.shape
    dx = torch.empty_like(x)
    L.check(L.lib().mi_gelu_gap_bwd(_p(x), _p(dout.contiguous().float()), _p(dx), B, Cc, H * W, _dt(x), _stream()), "gelu_gap_bwd")
    return dx


def ewise_fwd(a: Tensor, b: Tensor, op: int) -> Tensor:
    """op 0: a * b ; op 1: a * silu(b).  a, b: [B,C,H,W], channel slices allowed."""
    B = a.shape[0]
    Ln = a[0].numel()
    out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    L.check(L.lib().mi_ewise_fwd(_p(a), _bstride(a), _p(b), _bstride(b), _p(out), B, Ln, op, _dt(a), _stream()), "ewise_fwd")
    return out


def ewise_bwd(a: Tensor, b: Tensor, dout: Tensor, op: int, da: Optional[Tensor] = None, db: Optional[Tensor] = None):
    """Gradients of ewise_fwd; da / db may be given as (channel-slice) views to be written in place."""
    _gpu(dout)
    B = a.shape[0]
    Ln = a[0].numel()
    if da is None:
        da = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    if db is None:
        db = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    L.check(L.lib().mi_ewise_bwd(_p(a), _bstride(a), _p(b), _bstride(b), _p(dout), _p(da), _bstride(da), _p(db), _bstride(db), B,
                                 Ln, op, _dt(a), _stream()), "ewise_bwd")
    return da, db


# ----------------------------------------------------------------------------- training-step tail
def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 1e-2, grad_scale: float = 1.0, dev_scalars: Optional[Tensor] = None) -> None:
    """dev_scalars: optional device tensor [lr, 1-b1^t, sqrt(1-b2^t)] that overrides lr/step (graph replay)."""
    _gpu(p, g, m, v, dev_scalars)
    L.check(L.lib().mi_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step,
                                  grad_scale, _p(dev_scalars), _stream()), "adamw_step")


def l1_loss(a: Tensor, b: Tensor, want_grad: bool = True, scale: float = 1.0):
    """mean|a-b| and (optionally) its gradient w.r.t. a times ``scale``; loss returned as a 1-element fp32 tensor."""
    _gpu(a, b)
    buf = torch.zeros(1 + 1024, dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    n = a.numel()
    L.check(L.lib().mi_l1_loss(_p(a), _p(b), _p(da), _p(buf), n, scale / n, _dt(a), _stream()), "l1_loss")
    return buf[:1], da


# ----------------------------------------------------------------------------- profiler (bench.py roofline pass)
_pw_cache_buf: Optional[Tensor] = None


def pw_cache_enable(nbytes: int, device, params: Optional[Tensor] = None) -> Optional[Tensor]:
    """Lend the library a device buffer for packed 1x1 weights (include/mi_restore.h: mi_pw_cache_*).  `params` is the
    storage that holds the weights (e.g. the trainer's flat parameter buffer): only matrices inside it are cached.  The
    caller promises to call pw_cache_refresh() after every in-place weight update (or pw_cache_invalidate()).
    Returns the owner token (the lent buffer): hand it to pw_cache_release() - the cache is process-global, and only the
    object whose buffer it still points at may switch it off."""
    global _pw_cache_buf
    if nbytes <= 0 or params is None:
        L.check(L.lib().mi_pw_cache_enable(None, 0, None, None), "pw_cache_enable")
        _pw_cache_buf = None
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    lo = params.data_ptr()
    L.check(L.lib().mi_pw_cache_enable(buf.data_ptr(), nbytes, lo, lo + params.numel() * params.element_size()),
            "pw_cache_enable")
    _pw_cache_buf = buf  # keeps the memory alive for as long as the cache points at it
    return buf


def pw_cache_owner() -> Optional[Tensor]:
    return _pw_cache_buf


def pw_cache_release(token: Optional[Tensor]) -> bool:
    """Switch the cache off if (and only if) it still belongs to `token`'s owner.  A later owner (a second trainer, a
    PackedWeights made afterwards) keeps its cache when an earlier object is closed or garbage-collected."""
    global _pw_cache_buf
    if token is None or _pw_cache_buf is not token:
        return False
    L.check(L.lib().mi_pw_cache_enable(None, 0, None, None), "pw_cache_enable")
    _pw_cache_buf = None
    return True


def pw_cache_pending() -> bool:
    """True while the cache holds weights it has not packed yet (seen since the last refresh) or was invalidated."""
    return bool(L.lib().mi_pw_cache_pending())


def pw_cache_refresh() -> None:
    L.check(L.lib().mi_pw_cache_refresh(_stream()), "pw_cache_refresh")


def pw_cache_invalidate() -> None:
    L.check(L.lib().mi_pw_cache_invalidate(), "pw_cache_invalidate")


_deferred_arena: Optional[Tensor] = None


def deferred_begin(nbytes: int, device) -> Optional[Tensor]:
    """Lend the library an arena for deferred parameter-gradient reductions (include/mi_restore.h: mi_deferred_*): backward
    calls that accumulate into gradient buffers record their final sums instead of launching them; deferred_flush() runs them
    all in one launch.  Returns the owner token (the arena), or None when another owner already holds the context."""
    global _deferred_arena
    if _deferred_arena is not None:
        return None
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    L.check(L.lib().mi_deferred_begin(buf.data_ptr(), buf.numel()), "deferred_begin")
    _deferred_arena = buf
    return buf


def deferred_record(on: bool) -> None:
    """Producers defer only while recording is on: the owner brackets its backward window with it."""
    if _deferred_arena is not None:
        L.check(L.lib().mi_deferred_record(int(on)), "deferred_record")


def deferred_flush() -> None:
    if _deferred_arena is not None:
        L.check(L.lib().mi_deferred_flush(_stream()), "deferred_flush")


def deferred_pending() -> int:
    return int(L.lib().mi_deferred_pending()) if _deferred_arena is not None else 0


def deferred_end(token: Optional[Tensor]) -> bool:
    """Flush and close the deferral context if `token` owns it."""
    global _deferred_arena
    if token is None or _deferred_arena is not token:
        return False
    deferred_flush()
    deferred_record(False)
    L.check(L.lib().mi_deferred_end(), "deferred_end")
    _deferred_arena = None
    return True


def prof_enable(on: bool) -> None:
    L.check(L.lib().mi_prof_enable(int(on)), "prof_enable")


def prof_collect():
    """-> {kernel: dict(ms, bytes, flops, launches)} for kernels launched since prof_enable(True)."""
    lib = L.lib()
    n = lib.mi_prof_kernel_count()
    ms, by, fl = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    cnt = (L.c_i64 * n)()
    L.check(lib.mi_prof_collect(ms, by, fl, cnt, n), "prof_collect")
    out = {}
    for i in range(n):
        if cnt[i]:
            out[lib.mi_prof_kernel_name(i).decode()] = dict(ms=ms[i], bytes=by[i], flops=fl[i], launches=int(cnt[i]))
    return out


# ----------------------------------------------------------------------------- AdaIR frequency modules (csrc/adair.hip)
def box_down(img: Tensor, H: int, W: int) -> Tensor:
    """F.interpolate(img, (H, W), mode='bilinear') for the integer level factors (1, 2, 4, 8 ...)."""
    _gpu(img)
    B, Cc, Hi, Wi = img.shape
    if Hi % H or Wi % W or Hi // H != Wi // W:
        raise ValueError(f"box_down: {Hi}x{Wi} -> {H}x{W} is not an integer level factor")
    out = torch.empty((B, Cc, H, W), dtype=img.dtype, device=img.device)
    L.check(L.lib().mi_box_down(_p(img), _p(out), B, Cc, Hi, Wi, Hi // H, _dt(img), _stream()), "box_down")
    return out


def fre_rect(pooled: Tensor, w0: Tensor, w2: Tensor, H: int, W: int, n: int = 128) -> Tensor:
    _gpu(pooled, w0, w2)
    B, Cc = pooled.shape
    half = torch.empty((B, 2), dtype=torch.int32, device=pooled.device)
    L.check(L.lib().mi_fre_rect(_p(pooled), _p(w0), _p(w2), _p(half), B, Cc, w0.shape[0], H, W, n, _stream()), "fre_rect")
    return half


def fre_split_fwd(feat: Tensor, half: Optional[Tensor]):
    _gpu(feat, half)
    B, Cc, H, W = feat.shape
    high, low = torch.empty_like(feat), torch.empty_like(feat)
    coef = None
    if half is not None:
        coef = torch.empty(L.lib().mi_fre_split_coef_bytes(B, Cc) // 4, dtype=torch.float32, device=feat.device)
    L.check(L.lib().mi_fre_split_fwd(_p(feat), _p(half), _p(high), _p(low), _p(coef), B, Cc, H, W, _dt(feat), _stream()),
            "fre_split_fwd")
    return high, low, coef


def fre_split_bwd(feat: Tensor, half: Optional[Tensor], coef: Optional[Tensor], dhigh: Tensor, dlow: Tensor) -> Tensor:
    _gpu(feat, half, coef, dhigh, dlow)
    B, Cc, H, W = feat.shape
    dfeat = torch.empty_like(feat)
    ws = _ws(L.lib().mi_fre_split_workspace(B, Cc, H, W), feat.device)
    L.check(L.lib().mi_fre_split_bwd(_p(feat), _p(half), _p(coef), _p(dhigh), _p(dlow), _p(dfeat), B, Cc, H, W, _dt(feat),
                                     _p(ws), _stream()), "fre_split_bwd")
    return dfeat


def chan_maxmean_fwd(x: Tensor):
    _gpu(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, 2, H, W), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, H * W), dtype=torch.int32, device=x.device)
    L.check(L.lib().mi_chan_maxmean_fwd(_p(x), _p(out), _p(idx), B, Cc, H * W, _dt(x), _stream()), "chan_maxmean_fwd")
    return out, idx


def chan_maxmean_bwd(dout: Tensor, idx: Tensor, Cc: int) -> Tensor:
    _gpu(dout, idx)
    B, _, H, W = dout.shape
    dx = torch.empty((B, Cc, H, W), dtype=dout.dtype, device=dout.device)
    L.check(L.lib().mi_chan_maxmean_bwd(_p(dout), _p(idx), _p(dx), B, Cc, H * W, _dt(dout), _stream()), "chan_maxmean_bwd")
    return dx


def plane_max_fwd(x: Tensor):
    _gpu(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, Cc), dtype=torch.int32, device=x.device)
    L.check(L.lib().mi_plane_max_fwd(_p(x), _p(out), _p(idx), B * Cc, H * W, _dt(x), _stream()), "plane_max_fwd")
    return out, idx


def pool_pair_bwd(davg: Tensor, dmax: Tensor, idx: Tensor, like: Tensor) -> Tensor:
    _gpu(davg, dmax, idx)
    B, Cc, H, W = like.shape
    dx = torch.empty_like(like)
    L.check(L.lib().mi_pool_pair_bwd(_p(davg), _p(dmax), _p(idx), _p(dx), B * Cc, H * W, _dt(like), _stream()), "pool_pair_bwd")
    return dx


def chan_gate_fwd(avg: Tensor, mx: Tensor, w1: Tensor, w2: Tensor):
    _gpu(avg, mx, w1, w2)
    B, Cc = avg.shape
    R_ = w1.shape[0]
    cw = torch.empty_like(avg)
    hid = torch.empty((B, 2, R_), dtype=torch.float32, device=avg.device)
    L.check(L.lib().mi_chan_gate_fwd(_p(avg), _p(mx), _p(w1), _p(w2), _p(cw), _p(hid), B, Cc, R_, _stream()), "chan_gate_fwd")
    return cw, hid


def chan_gate_bwd(avg, mx, w1, w2, cw, hid, dcw, dw1, dw2, accumulate: bool):
    _gpu(avg, mx, w1, w2, cw, hid, dcw, dw1, dw2)
    B, Cc = avg.shape
    davg, dmx = torch.empty_like(avg), torch.empty_like(avg)
    L.check(L.lib().mi_chan_gate_bwd(_p(avg), _p(mx), _p(w1), _p(w2), _p(cw), _p(hid), _p(dcw), _p(davg), _p(dmx), _p(dw1), _p(dw2),
                                     B, Cc, w1.shape[0], int(accumulate), _stream()), "chan_gate_bwd")
    return davg, dmx


def refine_mix_fwd(low: Tensor, high: Tensor, s: Tensor, cw: Tensor) -> Tensor:
    _gpu(low, high, s, cw)
    B, Cc, H, W = low.shape
    out = torch.empty_like(low)
    L.check(L.lib().mi_refine_mix_fwd(_p(low), _p(high), _p(s), _p(cw), _p(out), B, Cc, H * W, _dt(low), _stream()), "refine_mix_fwd")
    return out


def refine_mix_bwd(low: Tensor, high: Tensor, s: Tensor, cw: Tensor, dout: Tensor):
    _gpu(low, high, s, cw, dout)
    B, Cc, H, W = low.shape
    dlow, dhigh, ds = torch.empty_like(low), torch.empty_like(low), torch.empty_like(s)
    dcw = torch.empty_like(cw)
    L.check(L.lib().mi_refine_mix_bwd(_p(low), _p(high), _p(s), _p(cw), _p(dout), _p(dlow), _p(dhigh), _p(ds), _p(dcw), B, Cc, H * W,
                                      _dt(low), _stream()), "refine_mix_bwd")
    return dlow, dhigh, ds, dcw


def scale_add_fwd(a: Tensor, y: Tensor, p1: Tensor, p2: Tensor) -> Tensor:
    _gpu(a, y, p1, p2)
    B, Cc, H, W = a.shape
    out = torch.empty_like(a)
    L.check(L.lib().mi_scale_add_fwd(_p(a), _p(y), _p(p1), _p(p2), _p(out), B, Cc, H * W, _dt(a), _stream()), "scale_add_fwd")
    return out


def scale_add_bwd(a: Tensor, y: Tensor, p1: Tensor, p2: Tensor, dout: Tensor, dp1: Tensor, dp2: Tensor, accumulate: bool):
    _gpu(a, y, p1, p2, dout, dp1, dp2)
    B, Cc, H, W = a.shape
    da, dy = torch.empty_like(a), torch.empty_like(a)
    L.check(L.lib().mi_scale_add_bwd(_p(a), _p(y), _p(p1), _p(p2), _p(dout), _p(da), _p(dy), _p(dp1), _p(dp2), B, Cc, H * W,
                                     int(accumulate), _dt(a), _stream()), "scale_add_bwd")
    return da, dy
