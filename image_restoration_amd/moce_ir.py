"""Drop-in replacement for the reference's top-level ``moce_ir.py`` (MoCE-IR): same class names, constructor arguments,
parameter names (``state_dict`` interchangeable) and ``forward`` signatures, rebuilt around the gfx950 kernels.

What runs where (reference line numbers are moce_ir.py of the upstream repository):
  * EncoderBlock / the MDTA, cross-MDTA (7x7 depthwise kv), GDFN and LayerNorm pieces of DecoderBlock: the block kernels of
    ``restormer.py`` (one autograd node per block).
  * RoutingFunction (:684-800): ONE launch forward, one backward (csrc/moce.hip): both linear gates, injected noise, softmax,
    top-k, gate scatter, the two CV^2 losses, and the dispatcher's sample -> expert tables.  The reference's nonzero / sort /
    split bookkeeping (:82-91) is not re-enacted: the tables come out of the router launch; the only host read-back is the E
    segment sizes (the ragged expert launches are enqueued from the host).
  * AdapterLayer (:584-681): rows gathered by the table (native gather), every expert runs on its contiguous segment, the
    experts' last projection (+ shortcut) is written by all experts into ONE stitched buffer (no concatenation), and the
    gate-weighted fp32 scatter-add puts the rows back.  Train and eval share this path; for batch 1 - the only case the
    reference's eval branch (:673-678) is meaningful for - the result is the same.
  * FFTAttention (:373-422): ``irfft2(rfft2(q) * rfft2(k))`` per patch is a 2-D circular convolution; it is computed as one
    (mi_patch_circconv, fp32 arithmetic like the reference's upcast) straight from the NCHW planes - no padding copy, no
    rearrange, no FFT library call.  k and v stay channel slices of the kv tensor.
  * FrequencyEmbedding (:1048-1075): the 3x3 high-pass depthwise conv is the native depthwise kernel, GELU + global mean one
    kernel, the two-layer MLP on [B, dim] the native pointwise GEMM (GELU between them as the gating kernel's sibling).
  * MoCEIR (:1080-1231): patch embedding / output conv / down / up-sampling are the native 3x3 kernels of ``restormer.py``;
    the decoder's ``cat -> 1x1`` fusion is a two-panel GEMM (no concatenated tensor).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .restormer import (Attention, Downsample, FeedForward, LayerNorm, OverlapPatchEmbed, Upsample, _apply,  # noqa: F401
                        _BlockFn, block_apply, _Conv1x1Fn, _conv1x1_module, _CrossAttentionFn, _DwConvFn, _conv2d, _grad_mode, _main_grads)

__all__ = ["SparseDispatcher", "LayerNorm", "FeedForward", "Attention", "CrossAttention", "FFTAttention", "MySequential",
           "ModExpert", "AdapterLayer", "RoutingFunction", "EncoderBlock", "DecoderBlock", "HighPassConv2d",
           "FrequencyEmbedding", "EncoderResidualGroup", "DecoderResidualGroup", "OverlapPatchEmbed", "Downsample", "Upsample",
           "MoCEIR"]

Tensor = torch.Tensor


def _c1(x, conv: nn.Conv2d):
    """1x1 conv module applied through the native pointwise GEMM."""
    return _apply(_Conv1x1Fn, x, None, conv.weight, conv.bias)


def _dw(x, conv: nn.Conv2d):
    return _apply(_DwConvFn, x, conv.weight, conv.bias)


def _acc_or_return(params, grads):
    """Parameter gradients either accumulate into the trainer's flat buffer (main_grad) or are returned to autograd."""
    mg = _main_grads(params)
    if mg is None:
        return list(grads)
    for m, g in zip(mg, grads):
        if m is not None and g is not None:
            m.add_(g)
    return [None] * len(grads)


# ======================================================================================
# autograd glue over csrc/moce.hip and csrc/dispatch.hip
# ======================================================================================
class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gap_fwd(x)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return ops.gap_bwd(dout, x)


class _GeluGapFn(torch.autograd.Function):
    """mean over the plane of gelu(x): FrequencyEmbedding's activation + pooling (:1062-1064,1071-1073) in one pass."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_gap_fwd(x)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return ops.gelu_gap_bwd(x, dout)


class _GeluFn(torch.autograd.Function):
    """erf-form GELU of a small tensor (FrequencyEmbedding's MLP activation, :1071) on the gating kernel (op 2)."""

    @staticmethod
    def forward(ctx, a):
        a4 = a.contiguous().view(1, 1, 1, -1)
        ctx.save_for_backward(a4)
        ctx.shape = a.shape
        return ops.ewise_fwd(a4, a4, 2).view(a.shape)

    @staticmethod
    def backward(ctx, dout):
        (a4,) = ctx.saved_tensors
        da, _ = ops.ewise_bwd(a4, a4, dout.contiguous().view(1, 1, 1, -1), 2, db=a4)    # db is not written for op 2
        return da.view(ctx.shape)


class _EwiseFn(torch.autograd.Function):
    """op 0: a * b ; op 1: a * silu(b)."""

    @staticmethod
    def forward(ctx, a, b, op):
        ctx.save_for_backward(a, b)
        ctx.op = op
        return ops.ewise_fwd(a, b, op)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da, db = ops.ewise_bwd(a, b, dout.contiguous(), ctx.op)
        return da, db, None


class _RowsGatherFn(torch.autograd.Function):
    """dispatch: out[i] = x[idx[i]] over whole feature maps; backward scatters the gradients back per sample."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n_rows = x.shape[0]
        return ops.rows_gather(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return ops.rows_scatter_add(dout.contiguous(), idx, None, ctx.n_rows, out_f32=False), None


class _RowsCombineFn(torch.autograd.Function):
    """combine: out[b] = sum_{i: idx[i]==b} gate[i] * src[i] in fp32 (:116-124)."""

    @staticmethod
    def forward(ctx, src, gates, idx, n_rows):
        src = src.contiguous()
        ctx.gshape = None if gates is None else tuple(gates.shape)
        gates = gates.reshape(-1).float().contiguous() if gates is not None else None
        ctx.save_for_backward(src, gates, idx)
        return ops.rows_scatter_add(src, idx, gates, n_rows, out_f32=True)

    @staticmethod
    def backward(ctx, dout):
        src, gates, idx = ctx.saved_tensors
        dout = dout.float().contiguous()
        dsrc = ops.rows_gather_scaled(dout, idx, gates, src.dtype)
        dgates = ops.rows_dot(dout, src, idx).reshape(ctx.gshape) if (gates is not None and ctx.needs_input_grad[1]) else None
        return dsrc, dgates, None, None


class _RouteFn(torch.autograd.Function):
    """The whole router in one launch each way.  Differentiable outputs: gates [B,E], top-k values [B,k], aux loss [1] and
    the gate of every dispatched row [B*k]; the index tables ride along as non-differentiable outputs."""

    @staticmethod
    def forward(ctx, pooled, freq, wg, wf, noise, complexity, k, training):
        gates, idx, vals, aux, tb = ops.moe_route_fwd(pooled, freq, wg, wf, noise, complexity, k, training)
        ctx.save_for_backward(pooled, freq, wg, wf, noise, idx, tb.logits, tb.row_of)
        ctx.complexity, ctx.training, ctx.tb = complexity, training, tb
        ctx.mg = _main_grads((wg, wf))
        for t in (idx, tb.counts, tb.offsets, tb.perm, tb.perm_expert, tb.row_of):
            ctx.mark_non_differentiable(t)
        return gates, idx, vals, aux, tb.perm_gate, tb.counts, tb.offsets, tb.perm, tb.perm_expert, tb.row_of

    @staticmethod
    def backward(ctx, dgates, _didx, dvals, daux, drow, *_):
        pooled, freq, wg, wf, noise, idx, logits, row_of = ctx.saved_tensors
        tb = ctx.tb
        if dvals is not None:      # gradient on the top-k values = gradient on the gate of the matching dispatched row
            extra = torch.zeros_like(tb.perm_gate).index_put_((row_of.reshape(-1).long(),), dvals.reshape(-1).float())
            drow = extra if drow is None else drow + extra
        dpooled, dfreq, dwg, dwf = ops.moe_route_bwd(pooled, freq, wg, wf, noise, ctx.complexity, tb, idx,
                                                      None if dgates is None else dgates.contiguous(),
                                                      None if drow is None else drow.contiguous(),
                                                      None if daux is None else daux.contiguous(), ctx.training)
        if ctx.mg is not None:
            ctx.mg[0].add_(dwg); ctx.mg[1].add_(dwf)
            dwg = dwf = None
        return dpooled, dfreq, dwg, dwf, None, None, None, None


class _FFTCoreFn(torch.autograd.Function):
    """LayerNorm(circconv_patch(q, k)) * v with k, v the two channel halves of ``kv`` (:402-419): three launches forward, the
    halves of d_kv written in place backward (no slice copies)."""

    @staticmethod
    def forward(ctx, q, kv, ln_w, ln_b, patch):
        c = q.shape[1]
        k, v = kv[:, :c], kv[:, c:]
        need = _grad_mode() and any(ctx.needs_input_grad)
        cc = ops.patch_circconv(q, k, patch)
        n, mean, rstd = ops.ln_fwd(cc, ln_w, ln_b, True, want_stats=need)
        out = ops.ewise_fwd(n, v, 0)
        if need:
            ctx.save_for_backward(q, kv, cc, n, mean, rstd, ln_w)
            ctx.patch = patch
            ctx.mg = _main_grads((ln_w, ln_b))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, cc, n, mean, rstd, ln_w = ctx.saved_tensors
        c = q.shape[1]
        k, v = kv[:, :c], kv[:, c:]
        dkv = torch.empty_like(kv)
        dn, _ = ops.ewise_bwd(n, v, dout.contiguous(), 0, db=dkv[:, c:])
        acc = ctx.mg is not None
        dw, db = ctx.mg if acc else (torch.empty_like(ln_w), torch.empty_like(ln_w))
        dcc = ops.ln_bwd(dn, cc, ln_w, mean, rstd, None, True, dw, db, acc)
        dq = ops.patch_circconv(dcc, k, ctx.patch, flip=True)
        ops.patch_circconv(dcc, q, ctx.patch, flip=True, out=dkv[:, :c])
        return dq, dkv, (None if acc else dw), (None if acc else db), None


def _w2(w: Tensor) -> Tensor:
    return w.reshape(w.shape[0], -1)


def _segment_rows(dev_offsets: Tensor, E: int, R: int) -> Tensor:
    """[E, R] int64: row (offsets[e] + i) of the stitched buffers for i = 0..R-1, clamped to the last row - the device-side
    replacement of the host slices ``rows[o:o + n]`` when the segment sizes are not read back (capacity mode)."""
    ar = torch.arange(R, device=dev_offsets.device, dtype=torch.int64)
    return (dev_offsets[:E].to(torch.int64).unsqueeze(1) + ar.unsqueeze(0)).clamp_(max=R - 1).contiguous()


class _ExpertsInFn(torch.autograd.Function):
    """First step of every expert at once (:552-555): a_e = proj[0]_e(x[rows of e]), g_e = proj[1]_e(shared[rows of e]) for all
    E experts in ONE grouped launch (csrc/grouped.hip) that reads the segment sizes / starts from the router's device tables.
    Backward: the two input gradients of all experts in one grouped launch (each expert writes its own row segment of the
    stitched gradient buffers); the weight gradients are per-expert Grams."""

    @staticmethod
    def forward(ctx, xrows, srows, counts, cap, dev_counts, dev_offsets, *ws):
        E = len(counts)
        w0, w1 = ws[:E], ws[E:]
        R, Cc, H, W = xrows.shape
        N = H * W
        # cap (capacity mode): counts[e] is every expert's buffer CAPACITY (all rows); the kernels write the true count of rows,
        # the rest stays zero so that the bodies see finite inputs and contribute nothing
        new = torch.zeros if cap else torch.empty
        outs_a = [new((n, w0[e].shape[0], H, W), dtype=xrows.dtype, device=xrows.device) for e, n in enumerate(counts)]
        outs_g = [new((n, w1[e].shape[0], H, W), dtype=xrows.dtype, device=xrows.device) for e, n in enumerate(counts)]
        probs = []
        for e, n in enumerate(counts):
            if n:
                probs.append(dict(x=xrows, w=_w2(w0[e]), y=outs_a[e], m=w0[e].shape[0], k=Cc, expert=e, y_local=True))
                probs.append(dict(x=srows, w=_w2(w1[e]), y=outs_g[e], m=w1[e].shape[0], k=Cc, expert=e, y_local=True))
        if probs:
            ops.grouped_pw_gemm(probs, dev_counts, dev_offsets, R, N, xrows.dtype)
        ctx.counts, ctx.cap = counts, cap
        ctx.save_for_backward(xrows, srows, dev_counts, dev_offsets, *ws)
        ctx.mg = [getattr(w, "main_grad", None) for w in ws]
        return tuple(outs_a) + tuple(outs_g)

    @staticmethod
    def backward(ctx, *douts):
        xrows, srows, dev_counts, dev_offsets, *ws = ctx.saved_tensors
        counts = ctx.counts
        E = len(counts)
        w0, w1 = ws[:E], ws[E:]
        R, Cc, H, W = xrows.shape
        da, dg = [d.contiguous() if d is not None else None for d in douts[:E]], [d.contiguous() if d is not None else None for d in douts[E:]]
        dx, ds = torch.empty_like(xrows), torch.empty_like(srows)
        probs = []
        for e, n in enumerate(counts):
            if not n:
                continue
            if da[e] is None:
                da[e] = torch.zeros((n, w0[e].shape[0], H, W), dtype=xrows.dtype, device=xrows.device)
            if dg[e] is None:
                dg[e] = torch.zeros((n, w1[e].shape[0], H, W), dtype=xrows.dtype, device=xrows.device)
            probs.append(dict(x=da[e], w=_w2(w0[e]), transposed=True, y=dx, m=Cc, k=w0[e].shape[0], expert=e, x_local=True))
            probs.append(dict(x=dg[e], w=_w2(w1[e]), transposed=True, y=ds, m=Cc, k=w1[e].shape[0], expert=e, x_local=True))
        if probs:
            ops.grouped_pw_gemm(probs, dev_counts, dev_offsets, R, H * W, xrows.dtype)
        dws = []
        o = 0
        idx = _segment_rows(dev_offsets, E, R) if ctx.cap else None
        for which, (d_list, src) in enumerate(((da, xrows), (dg, srows))):
            o = 0
            for e, n in enumerate(counts):
                w = ws[which * E + e]
                if not n:
                    dws.append(None)
                    continue
                # the expert's rows of the stitched source: a host slice, or (capacity mode) a gather by the device-side
                # segment start - rows past the segment meet zero gradient rows
                rows_e = ops.rows_gather(src, idx[e]) if ctx.cap else src[o:o + n]
                mg = ctx.mg[which * E + e]
                if mg is not None:                       # straight into the trainer's gradient slot (deferrable final sum)
                    ops.gram(d_list[e], rows_e, 1, True, out=mg, accumulate=True)
                    g = None
                else:
                    g = ops.gram(d_list[e], rows_e, 1, True)[0].reshape(w.shape)
                dws.append(g)
                o += n
        return (dx, ds, None, None, None, None) + tuple(dws)


class _ExpertsOutFn(torch.autograd.Function):
    """Last step of every expert at once (:556-558): out[rows of e] = proj[2]_e(t_e) + x[rows of e], written by ONE grouped
    launch into ONE row-stitched buffer (the reference concatenates the experts' outputs afterwards, :116); the input
    gradients dt_e = proj[2]_e^T dout[rows of e] of all experts are one grouped launch too."""

    @staticmethod
    def forward(ctx, xrows, counts, cap, dev_counts, dev_offsets, *tw):
        E = len(counts)
        ts, ws = tw[:E], tw[E:]
        out = torch.empty_like(xrows)
        R, Cc, H, W = xrows.shape
        probs = [dict(x=ts[e], w=_w2(ws[e]), r=xrows, y=out, m=Cc, k=ws[e].shape[1], expert=e, x_local=True)
                 for e, n in enumerate(counts) if n]
        if probs:
            ops.grouped_pw_gemm(probs, dev_counts, dev_offsets, R, H * W, xrows.dtype)
        ctx.counts, ctx.cap = counts, cap
        ctx.save_for_backward(dev_counts, dev_offsets, *[t for t in ts if t is not None], *ws)
        ctx.present = [t is not None for t in ts]
        ctx.mg = [getattr(w, "main_grad", None) for w in ws]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        E = len(ctx.counts)
        dev_counts, dev_offsets, *saved = ctx.saved_tensors
        nt = sum(ctx.present)
        it = iter(saved[:nt])
        ts = [next(it) if pr else None for pr in ctx.present]
        ws = saved[nt:]
        R, Cc, H, W = dout.shape
        dts = [(torch.zeros_like(t) if ctx.cap else torch.empty_like(t)) if t is not None else None for t in ts]
        probs = [dict(x=dout, w=_w2(ws[e]), transposed=True, y=dts[e], m=ws[e].shape[1], k=Cc, expert=e, y_local=True)
                 for e, n in enumerate(ctx.counts) if n]
        if probs:
            ops.grouped_pw_gemm(probs, dev_counts, dev_offsets, R, H * W, dout.dtype)
        dws = []
        o = 0
        idx = _segment_rows(dev_offsets, E, R) if ctx.cap else None
        for e, n in enumerate(ctx.counts):
            if not n:
                dws.append(None)
                continue
            rows_e = ops.rows_gather(dout, idx[e]) if ctx.cap else dout[o:o + n]      # (rows past the segment meet ts[e] rows that are zero)
            if ctx.mg[e] is not None:
                ops.gram(rows_e, ts[e], 1, True, out=ctx.mg[e], accumulate=True)
                g = None
            else:
                g = ops.gram(rows_e, ts[e], 1, True)[0].reshape(ws[e].shape)
            dws.append(g)
            o += n
        return (dout, None, None, None, None) + tuple(dts) + tuple(dws)


# ======================================================================================
# modules (reference interface)
# ======================================================================================
class SparseDispatcher(object):
    """The reference's helper API (:71-143) - ``dispatch`` / ``combine`` / ``expert_to_gates`` over gates [B, E] - kept for
    callers that use it directly.  (AdapterLayer does not: its tables come out of the router launch.)  The index is the list of
    (expert, sample) pairs with a non-zero gate in expert-major order - one ``nonzero`` of the transposed mask."""

    def __init__(self, num_experts, gates):
        self._gates = gates
        self._num_experts = num_experts
        pairs = torch.nonzero(gates.t() > 0)                    # rows sorted by expert, then by sample
        self._expert_index = pairs[:, :1]
        self._batch_index = pairs[:, 1].contiguous()
        self._part_sizes = torch.bincount(pairs[:, 0], minlength=num_experts).tolist()      # host read-back, as :88
        self._nonzero_gates = gates[self._batch_index, pairs[:, 0]].unsqueeze(1)

    def dispatch(self, inp):
        if inp.is_cuda and inp.dim() == 4 and inp.dtype in (torch.float32, torch.bfloat16):
            rows = _RowsGatherFn.apply(inp, self._batch_index)  # native row gather (csrc/dispatch.hip)
        else:
            rows = inp[self._batch_index]
        return torch.split(rows, self._part_sizes, dim=0)

    def combine(self, expert_out, multiply_by_gates=True):
        stitched = torch.cat(expert_out, 0)
        if stitched.is_cuda and stitched.dim() == 4 and stitched.dtype in (torch.float32, torch.bfloat16):
            return _RowsCombineFn.apply(stitched, self._nonzero_gates if multiply_by_gates else None, self._batch_index,
                                        self._gates.size(0))
        if multiply_by_gates:
            stitched = stitched * self._nonzero_gates.view(-1, 1, 1, 1)
        out = torch.zeros((self._gates.size(0),) + tuple(stitched.shape[1:]), dtype=torch.float32, device=stitched.device)
        return out.index_add(0, self._batch_index, stitched.float())

    def expert_to_gates(self):
        return torch.split(self._nonzero_gates, self._part_sizes, dim=0)


class CrossAttention(nn.Module):
    """MDTA with q from ``x`` (dw 3x3) and k, v from ``y`` (dw 7x7)  (:325-368)."""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=bias)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=7, stride=1, padding=7 // 2, groups=dim * 2, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def _params(self):
        return (self.temperature, self.q.weight, self.q.bias, self.q_dwconv.weight, self.q_dwconv.bias, self.kv.weight,
                self.kv.bias, self.kv_dwconv.weight, self.kv_dwconv.bias, self.project_out.weight, self.project_out.bias)

    def forward(self, x, y):
        return _apply(_CrossAttentionFn, x, y, self.num_heads, *self._params())


class FFTAttention(nn.Module):
    """Expert body (:373-422): per-patch circular convolution of q and k, LayerNorm, gate by v, 1x1 out."""

    def __init__(self, dim: int, **kwargs):
        super().__init__()
        self.patch_size = kwargs["patch_size"]
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=False)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=False)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=7, stride=1, padding=7 // 2, groups=dim * 2)
        self.norm = LayerNorm(dim, "WithBias")
        self.proj_out = nn.Conv2d(dim, dim, kernel_size=1, padding=0)

    def forward(self, x):
        q = _dw(_c1(x, self.q), self.q_dwconv)
        kv = _dw(_c1(x, self.kv), self.kv_dwconv)
        if self.patch_size not in (4, 8, 16, 32):
            # the reference configurations use 2 ** (i + 2), i < 4 (moce_ir.py:612); other sizes have no native kernel and
            # there is no FFT-library path
            raise NotImplementedError(f"FFTAttention: patch_size {self.patch_size} has no native kernel (built: 4, 8, 16, 32)")
        core = _apply(_FFTCoreFn, q, kv, self.norm.body.weight, self.norm.body.bias, self.patch_size)
        return _c1(core, self.proj_out)


class MySequential(nn.Sequential):
    """nn.Sequential whose layers take (x1, x2) (:31-50)."""

    def forward(self, x1, x2):
        for layer in self:
            x1 = layer(x1, x2)
        return x1


class ModExpert(nn.Module):
    """Low-rank expert: proj[0] C->r, body, gate by silu(proj[1](shared)), proj[2] r->C, + shortcut (:520-579)."""

    def __init__(self, dim: int, rank: int, func: nn.Module, depth: int, patch_size: int, kernel_size: int):
        super().__init__()
        self.depth = depth
        self.proj = nn.ModuleList([
            nn.Conv2d(dim, rank, kernel_size=1, padding=0, bias=False),
            nn.Conv2d(dim, rank, kernel_size=1, padding=0, bias=False),
            nn.Conv2d(rank, dim, kernel_size=1, padding=0, bias=False)
        ])
        self.body = func(rank, kernel_size=kernel_size, patch_size=patch_size)

    def inner(self, x, shared):
        """Everything before the last projection: body(proj[0] x) * silu(proj[1] shared)  -> [b, rank, H, W]."""
        return _apply(_EwiseFn, self.body(_c1(x, self.proj[0])), _c1(shared, self.proj[1]), 1)

    def process(self, x, shared):
        t = self.inner(x, shared)
        n = x.shape[0]
        cnt = torch.tensor([n], dtype=torch.int32, device=x.device)
        off = torch.zeros(1, dtype=torch.int32, device=x.device)
        return _apply(_ExpertsOutFn, x, (n,), False, cnt, off, t, self.proj[2].weight)      # proj[2](t) + x in one GEMM epilogue

    def forward(self, x, shared):
        if x.shape[0] == 0:
            return x
        # the reference re-applies `process` to the SAME input `depth` times and keeps the last result (:567-570):
        # depth only repeats identical work, so it is applied once
        return self.process(x.contiguous(), shared.contiguous())


class RoutingFunction(nn.Module):
    """Noisy top-k router (:684-800).  gate = GAP -> Linear(dim, E); + Linear(freq_dim, E)(freq_emb)."""

    def __init__(self, dim, freq_dim, num_experts, k, complexity, use_complexity_bias: bool = True,
                 complexity_scale: str = "max"):
        super().__init__()
        # indices 0/1 hold no parameters; they only keep the reference's key 'gate.2.weight'
        self.gate = nn.Sequential(nn.Identity(), nn.Identity(), nn.Linear(dim, num_experts, bias=False))
        self.freq_gate = nn.Linear(freq_dim, num_experts, bias=False)
        if complexity_scale == "min":
            complexity = complexity / complexity.min()
        elif complexity_scale == "max":
            complexity = complexity / complexity.max()
        self.register_buffer('complexity', complexity)
        self.k = k
        self.tau = 1
        self.num_experts = num_experts
        self.noise_std = (1.0 / num_experts) * 1.0
        self.use_complexity_bias = use_complexity_bias
        self.tables = None

    def route(self, x, freq_emb):
        """-> (gates, top-k indices, top-k values, aux [1], gate of every dispatched row, index tables)."""
        pooled = _GapFn.apply(x.contiguous())                                 # native GAP, fp32 [B, C]
        noise = torch.randn_like(pooled[:, :1].expand(-1, self.num_experts))   # N(0,1) draw, train AND eval (:741)
        cx = self.complexity.float() * self.tau if self.use_complexity_bias else None
        out = _apply(_RouteFn, pooled, freq_emb.float().contiguous(), self.gate[2].weight, self.freq_gate.weight,
                     noise.float().contiguous(), cx, self.k, self.training)
        gates, idx, vals, aux, rowg = out[:5]
        tb = ops.RouteTables()
        tb.counts, tb.offsets, tb.perm, tb.perm_expert, tb.row_of = out[5:]
        tb.perm_gate = rowg
        self.tables = tb
        return gates, idx, vals, aux, rowg, tb

    def forward(self, x, freq_emb):
        gates, idx, vals, aux, _, _ = self.route(x, freq_emb)
        return gates, idx, vals, (aux[0] if self.training else 0)


def _ladder(kind, base, n, dim=None):
    """Per-expert depth / rank schedules (:616-644)."""
    if isinstance(kind, int):
        return [kind] * n
    steps = {"lin": lambda i: base + i, "double": lambda i: base + 2 * i, "constant": lambda i: base}
    if kind in steps:
        return [steps[kind](i) for i in range(n)]
    return None


class AdapterLayer(nn.Module):
    """E experts of growing rank / patch / kernel behind the noisy top-k router, then a 1x1 projection (:584-681)."""

    CAPACITY_ROWS = 32                  # B * k up to which every expert runs over all rows (capacity mode, no host read-back)

    def __init__(self, dim: int, rank: int, num_experts: int = 4, top_k: int = 2, expert_layer: nn.Module = FFTAttention,
                 stage_depth: int = 1, depth_type: str = "lin", rank_type: str = "constant", freq_dim: int = 128,
                 with_complexity: bool = False, complexity_scale: str = "min"):
        super().__init__()
        self.tau = 1
        self.loss = None
        self.top_k = top_k
        self.noise_eps = 1e-2
        self.num_experts = num_experts
        self.dispatch = "ragged"        # "ragged" | "capacity" | "auto" (see forward); MI_MOCE_DISPATCH overrides
        E = num_experts
        depths = _ladder(depth_type, stage_depth, E)
        if depths is None:
            depths = {"exp": [2 ** i for i in range(E)], "fact": [math.factorial(i + 1) for i in range(E)]}.get(depth_type)
        ranks = _ladder(rank_type, rank, E)
        if ranks is None:
            ranks = {"exp": [rank ** (i + 1) for i in range(E)], "fact": [math.factorial(rank + i) for i in range(E)],
                     "spread": [dim // (2 ** (E - 1 - i)) for i in range(E)]}.get(rank_type)
        if depths is None or ranks is None or isinstance(rank_type, int):
            raise NotImplementedError
        self.experts = nn.ModuleList([
            MySequential(ModExpert(dim, rank=ranks[i], func=expert_layer, depth=depths[i], patch_size=2 ** (i + 2),
                                   kernel_size=3 + 2 * i)) for i in range(E)])
        self.proj_out = nn.Conv2d(dim, dim, kernel_size=1, padding=0, bias=False)
        expert_complexity = torch.tensor([sum(p.numel() for p in expert.parameters()) for expert in self.experts])
        self.routing = RoutingFunction(dim, freq_dim, num_experts=E, k=top_k, complexity=expert_complexity,
                                       use_complexity_bias=with_complexity, complexity_scale=complexity_scale)

    def forward(self, x, freq_emb, shared):
        gates, idx, vals, aux, row_gate, tb = self.routing.route(x, freq_emb)
        self.loss = aux[0] if self.training else 0
        # Segment sizes.  Ragged mode (default): the E sizes come to the host (as the reference's .tolist(), :88) so that the
        # bodies run on exactly their rows.  Capacity mode (dispatch = "capacity", or "auto" with B * k <= CAPACITY_ROWS):
        # NOTHING is read back - every expert's buffers hold all B * k rows, the grouped launches write / read the true number
        # of rows (device tables) and the bodies run over zero rows beyond it; static shapes, no host sync, so the whole step
        # can be captured in a HIP graph (bench.py --model moce --graph 1).  Measured at bs 8: the E-fold body work
        # and the per-expert row gathers of the weight gradients cost more than the 20 small read-backs they remove, and the
        # step is bound by the GPU's ~15 us per dependent tiny kernel either way - hence not the default.
        E = self.num_experts
        R = x.shape[0] * self.top_k
        mode = ops.env("MI_MOCE_DISPATCH") or self.dispatch
        cap = mode == "capacity" or (mode == "auto" and R <= self.CAPACITY_ROWS)
        counts = (R,) * E if cap else tuple(tb.counts.tolist())
        xrows = _RowsGatherFn.apply(x.contiguous(), tb.perm)
        srows = _RowsGatherFn.apply(shared.contiguous(), tb.perm)
        mods = [ex[0] for ex in self.experts]
        ag = _apply(_ExpertsInFn, xrows, srows, counts, cap, tb.counts, tb.offsets, *[m_.proj[0].weight for m_ in mods],
                    *[m_.proj[1].weight for m_ in mods])
        inner: List[Optional[Tensor]] = [
            _apply(_EwiseFn, mods[e].body(ag[e]), ag[E + e], 1) if counts[e] else None for e in range(E)]
        rows_out = _apply(_ExpertsOutFn, xrows, counts, cap, tb.counts, tb.offsets, *inner, *[m_.proj[2].weight for m_ in mods])
        out = _RowsCombineFn.apply(rows_out, row_gate, tb.perm, x.shape[0])     # gate multiply + fp32 scatter-add (:116-124)
        return _c1(out.to(x.dtype), self.proj_out)


class EncoderBlock(nn.Module):
    """x + mixer(norms[0](x)); + ffn(norms[1](.))  (:805-834): the Restormer block under MoCE's names."""

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norms = nn.ModuleList([LayerNorm(dim, LayerNorm_type), LayerNorm(dim, LayerNorm_type)])
        self.mixer = Attention(dim, num_heads, bias)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x):
        params = self.norms[0]._params() + self.mixer._params() + self.norms[1]._params() + self.ffn._params()
        return block_apply(x, self.mixer.num_heads, params)


class DecoderBlock(nn.Module):
    """Shared MDTA + MoCE adapter + cross-MDTA mixer + GDFN (:839-897).  Returns (x, adapter.loss)."""

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type, expert_layer, complexity_scale=None,
                 rank=None, num_experts=None, top_k=None, depth_type=None, rank_type=None, stage_depth=None,
                 freq_dim: int = 128, with_complexity: bool = False):
        super().__init__()
        self.norms = nn.ModuleList([LayerNorm(dim, LayerNorm_type), LayerNorm(dim, LayerNorm_type)])
        self.proj = nn.ModuleList([nn.Conv2d(dim, dim, kernel_size=1, padding=0), nn.Conv2d(dim, dim, kernel_size=1, padding=0)])
        self.shared = Attention(dim, num_heads, bias)
        self.mixer = CrossAttention(dim, num_heads=num_heads, bias=bias)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)
        self.adapter = AdapterLayer(dim, rank, top_k=top_k, num_experts=num_experts, expert_layer=expert_layer,
                                    freq_dim=freq_dim, depth_type=depth_type, rank_type=rank_type, stage_depth=stage_depth,
                                    with_complexity=with_complexity, complexity_scale=complexity_scale)

    def forward(self, x, freq_emb=None):
        shortcut = x
        x = self.norms[0](x)
        x_s = _c1(x, self.proj[0])
        x_a = _c1(x, self.proj[1])
        x_s = self.shared(x_s)
        x_a = self.adapter(x_a, freq_emb, x_s)
        x = self.mixer(x_a, x_s) + shortcut
        x = x + self.ffn(self.norms[1](x))
        return x, self.adapter.loss


class HighPassConv2d(nn.Module):
    """Depthwise 3x3 initialised to the 8-neighbour Laplacian (:224-249).  (The reference's ``freeze`` flag assigns an
    attribute instead of calling ``requires_grad_``, so the kernel stays trainable there - and here.)"""

    def __init__(self, c, freeze):
        super().__init__()
        self.conv = nn.Conv2d(in_channels=c, out_channels=c, kernel_size=3, padding=1, bias=False, groups=c)
        lap = -torch.ones(3, 3)
        lap[1, 1] = 8.0
        with torch.no_grad():
            self.conv.weight.copy_(lap.expand(c, 1, 3, 3))

    def forward(self, x):
        return _dw(x, self.conv)


def _linear_rows(x: Tensor, lin: nn.Linear) -> Tensor:
    """nn.Linear on [B, D] rows through the pointwise GEMM: the rows become the pixels of a one-image plane."""
    xt = x.t().contiguous().view(1, x.shape[1], 1, x.shape[0])
    y = _apply(_Conv1x1Fn, xt, None, lin.weight.view(lin.out_features, lin.in_features, 1, 1), lin.bias)
    return y.view(lin.out_features, x.shape[0]).t()


class FrequencyEmbedding(nn.Module):
    """High-pass -> GELU -> global mean -> Linear / GELU / Linear on the latent features (:1048-1075)."""

    def __init__(self, dim):
        super().__init__()
        self.high_conv = nn.Sequential(HighPassConv2d(dim, freeze=True), nn.GELU())
        self.mlp = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))

    def forward(self, x):
        pooled = _GeluGapFn.apply(self.high_conv[0](x).contiguous())          # [B, dim] fp32 ; GELU + mean in one pass
        h = _linear_rows(pooled, self.mlp[0])
        return _linear_rows(_GeluFn.apply(h), self.mlp[2])


class EncoderResidualGroup(nn.Module):
    """``num_blocks`` EncoderBlocks in a row (:926-958)."""

    def __init__(self, dim: int, num_heads: List[int], num_blocks: int, ffn_expansion: int, LayerNorm_type: str, bias: bool):
        super().__init__()
        self.loss = None
        self.num_blocks = num_blocks
        self.layers = nn.ModuleList([EncoderBlock(dim, num_heads, ffn_expansion, bias, LayerNorm_type)
                                     for _ in range(num_blocks)])

    def forward(self, x):
        self.loss = 0
        for blk in self.layers:
            x = blk(x)
        return x


class DecoderResidualGroup(nn.Module):
    """``num_blocks`` DecoderBlocks in a row; ``loss`` sums their routers' auxiliary losses (:962-1013)."""

    def __init__(self, dim: int, num_heads: List[int], num_blocks: int, ffn_expansion: int, LayerNorm_type: str, bias: bool,
                 complexity_scale=None, rank=None, num_experts=None, expert_layer=None, top_k=None, depth_type=None,
                 stage_depth=None, rank_type=None, freq_dim: int = 128, with_complexity: bool = False):
        super().__init__()
        self.loss = None
        self.num_blocks = num_blocks
        self.layers = nn.ModuleList([
            DecoderBlock(dim, num_heads, ffn_expansion, bias, LayerNorm_type, expert_layer=expert_layer, rank=rank,
                         num_experts=num_experts, top_k=top_k, stage_depth=stage_depth, freq_dim=freq_dim,
                         complexity_scale=complexity_scale, depth_type=depth_type, rank_type=rank_type,
                         with_complexity=with_complexity) for _ in range(num_blocks)])

    def forward(self, x, freq_emb=None):
        total = 0
        for blk in self.layers:
            x, aux = blk(x, freq_emb)
            total = total + aux
        self.loss = total
        return x


class MoCEIR(nn.Module):
    """The MoCE-IR U-Net (:1080-1231): encoder groups -> latent -> frequency embedding -> MoCE decoder groups -> refinement
    -> 3x3 output + input.  ``total_loss`` holds the mean auxiliary (load-balance) loss of the decoder blocks after a forward."""

    def __init__(self, inp_channels=3, out_channels=3, dim=32, levels: int = 4, heads=[1, 1, 1, 1], num_blocks=[1, 1, 1, 3],
                 num_dec_blocks=[1, 1, 1], ffn_expansion_factor=2, num_refinement_blocks=1, LayerNorm_type='WithBias',
                 bias=False, rank=2, num_experts=4, depth_type="lin", stage_depth=[3, 2, 1], rank_type="constant", topk=1,
                 expert_layer=FFTAttention, with_complexity=False, complexity_scale="max"):
        super().__init__()
        self.levels = levels
        self.num_blocks = num_blocks
        self.num_dec_blocks = num_dec_blocks
        self.num_refinement_blocks = num_refinement_blocks
        widths = [dim << i for i in range(levels)]
        self.patch_embed = OverlapPatchEmbed(in_c=inp_channels, embed_dim=dim, bias=False)
        self.freq_embed = FrequencyEmbedding(widths[-1])
        self.enc = nn.ModuleList([
            nn.ModuleList([EncoderResidualGroup(dim=widths[i], num_blocks=num_blocks[i], num_heads=heads[i],
                                                ffn_expansion=ffn_expansion_factor, LayerNorm_type=LayerNorm_type, bias=True),
                           Downsample(widths[i])]) for i in range(levels - 1)])
        self.latent = EncoderResidualGroup(dim=widths[-1], num_blocks=num_blocks[-1], num_heads=heads[-1],
                                           ffn_expansion=ffn_expansion_factor, LayerNorm_type=LayerNorm_type, bias=True)
        self.dec = nn.ModuleList([])
        for j in range(levels - 1):                   # decoder stage j works at level (levels - 2 - j)
            lvl = levels - 2 - j
            self.dec.append(nn.ModuleList([
                Upsample(widths[lvl + 1]),
                nn.Conv2d(widths[lvl + 1], widths[lvl], kernel_size=1, bias=bias),
                DecoderResidualGroup(dim=widths[lvl], num_blocks=num_dec_blocks[levels - 2 - j], num_heads=heads[lvl],
                                     ffn_expansion=ffn_expansion_factor, LayerNorm_type=LayerNorm_type, bias=bias,
                                     expert_layer=expert_layer, freq_dim=widths[-1], with_complexity=with_complexity, rank=rank,
                                     num_experts=num_experts, stage_depth=stage_depth[j], depth_type=depth_type,
                                     rank_type=rank_type, top_k=topk, complexity_scale=complexity_scale)]))
        self.refinement = EncoderResidualGroup(dim=dim, num_blocks=num_refinement_blocks, num_heads=heads[0],
                                               ffn_expansion=ffn_expansion_factor, LayerNorm_type=LayerNorm_type, bias=True)
        self.output = nn.Conv2d(dim, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.total_loss = None

    def forward(self, x, labels=None):
        feats = self.patch_embed(x)
        skips = []
        for group, down in self.enc:
            feats = group(feats)
            skips.append(feats)
            feats = down(feats)
        feats = self.latent(feats)
        freq_emb = self.freq_embed(feats)
        aux = 0
        for up, fuse, group in self.dec:
            # fusion(cat([up(feats), skip])) as ONE two-panel 1x1 GEMM: the concatenated tensor never exists (:1222)
            feats = _conv1x1_module(up(feats), skips.pop(), fuse)
            feats = group(feats, freq_emb)
            aux = aux + group.loss
        feats = self.refinement(feats)
        out = _conv2d(feats, self.output, x)          # 3x3 output conv with the input residual in its epilogue (:1228)
        self.total_loss = aux / sum(self.num_dec_blocks)
        return out
