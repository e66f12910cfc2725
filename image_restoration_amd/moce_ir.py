"""Drop-in replacements for the block classes of the reference's top-level ``moce_ir.py`` (MoCE-IR).

Same class names, constructor arguments, parameter names and ``forward`` signatures; the heavy parts run on the gfx950
kernels: channel LayerNorm, MDTA (``Attention``), cross-MDTA with the 7x7 depthwise kv branch (``CrossAttention``),
GDFN (``FeedForward``), every 1x1 projection of ``DecoderBlock`` / ``ModExpert`` / ``AdapterLayer`` and the router's
global average pool.  Still PyTorch-ROCm ops this round ("next"): the patch-FFT correlation inside
``FFTAttention`` and the [B, E] scalar math of the router (softmax / top-k / CV^2 losses).

``AdapterLayer`` keeps the reference's data flow (per-expert ragged sub-batches through ``SparseDispatcher``, one host
sync per call as in moce_ir.py:88); the dispatcher's gather and gate-weighted scatter-add run as native row kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions.normal import Normal

from . import ops
from .restormer import (Attention, FeedForward, LayerNorm, _apply, _BlockFn, _Conv1x1Fn, _CrossAttentionFn,  # noqa: F401
                        _DwConvFn)

__all__ = ["SparseDispatcher", "LayerNorm", "FeedForward", "Attention", "CrossAttention", "FFTAttention", "MySequential",
           "ModExpert", "AdapterLayer", "RoutingFunction", "EncoderBlock", "DecoderBlock"]


def _c1(x, conv: nn.Conv2d):
    """1x1 conv module applied through the native pointwise GEMM."""
    return _apply(_Conv1x1Fn, x, None, conv.weight, conv.bias)


def _dw(x, conv: nn.Conv2d):
    return _apply(_DwConvFn, x, conv.weight, conv.bias)


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gap_fwd(x)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return ops.gap_bwd(dout, x)


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
    """combine: out[b] = sum_{i: idx[i]==b} gate[i] * src[i] in fp32 (moce_ir.py:116-124)."""

    @staticmethod
    def forward(ctx, src, gates, idx, n_rows):
        src = src.contiguous()
        gates = gates.reshape(-1).float().contiguous() if gates is not None else None
        ctx.save_for_backward(src, gates, idx)
        return ops.rows_scatter_add(src, idx, gates, n_rows, out_f32=True)

    @staticmethod
    def backward(ctx, dout):
        src, gates, idx = ctx.saved_tensors
        dout = dout.float().contiguous()
        dsrc = ops.rows_gather_scaled(dout, idx, gates, src.dtype)
        dgates = ops.rows_dot(dout, src, idx).reshape(-1, 1) if (gates is not None and ctx.needs_input_grad[1]) else None
        return dsrc, dgates, None, None


class SparseDispatcher(object):
    """Sample -> expert bookkeeping with the reference's API (moce_ir.py:71-143): ``dispatch`` gathers the rows of the
    batch routed to each expert, ``combine`` scatters the gate-weighted expert outputs back (fp32 accumulation)."""

    def __init__(self, num_experts, gates):
        self._gates = gates
        self._num_experts = num_experts
        nz = torch.nonzero(gates)
        sorted_experts, index_sorted_experts = nz.sort(0)
        _, self._expert_index = sorted_experts.split(1, dim=1)
        self._batch_index = nz[index_sorted_experts[:, 1], 0]
        self._part_sizes = (gates > 0).sum(0).tolist()          # host sync, as moce_ir.py:88
        gates_exp = gates[self._batch_index.flatten()]
        self._nonzero_gates = torch.gather(gates_exp, 1, self._expert_index)

    def dispatch(self, inp):
        idx = self._batch_index.flatten()
        if inp.is_cuda and inp.dim() == 4 and inp.dtype in (torch.float32, torch.bfloat16):
            inp_exp = _RowsGatherFn.apply(inp, idx)                      # native row gather (csrc/dispatch.hip)
        else:
            inp_exp = inp[self._batch_index].squeeze(1)
        return torch.split(inp_exp, self._part_sizes, dim=0)

    def combine(self, expert_out, multiply_by_gates=True):
        stitched = torch.cat(expert_out, 0)
        if stitched.is_cuda and stitched.dim() == 4 and stitched.dtype in (torch.float32, torch.bfloat16):
            # gate multiply + index_add into fp32 zeros in one deterministic scatter (moce_ir.py:118-124)
            return _RowsCombineFn.apply(stitched, self._nonzero_gates if multiply_by_gates else None,
                                        self._batch_index.flatten(), self._gates.size(0))
        if multiply_by_gates:
            stitched = stitched.mul(self._nonzero_gates.unsqueeze(-1).unsqueeze(-1))
        zeros = torch.zeros(self._gates.size(0), expert_out[-1].size(1), expert_out[-1].size(2), expert_out[-1].size(3),
                            requires_grad=True, device=stitched.device)
        return zeros.index_add(0, self._batch_index, stitched.float())

    def expert_to_gates(self):
        return torch.split(self._nonzero_gates, self._part_sizes, dim=0)


class CrossAttention(nn.Module):
    """MDTA with q from ``x`` (dw 3x3) and k, v from ``y`` (dw 7x7)  (moce_ir.py:325-368)."""

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
    """Expert body (moce_ir.py:373-422): per-patch circular correlation of q and k through rfft2, LayerNorm, gate by v."""

    def __init__(self, dim: int, **kwargs):
        super().__init__()
        self.patch_size = kwargs["patch_size"]
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=False)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=False)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=7, stride=1, padding=7 // 2, groups=dim * 2)
        self.norm = LayerNorm(dim, "WithBias")
        self.proj_out = nn.Conv2d(dim, dim, kernel_size=1, padding=0)

    def pad_and_rearrange(self, x):
        b, c, h, w = x.shape
        p = self.patch_size
        pad_h, pad_w = (p - (h % p)) % p, (p - (w % p)) % p
        x = F.pad(x, (0, pad_w, 0, pad_h), mode='constant', value=0)
        hh, ww = x.shape[-2] // p, x.shape[-1] // p
        return x.reshape(b, c, hh, p, ww, p).permute(0, 1, 2, 4, 3, 5)          # b c h w p1 p2

    def rearrange_to_original(self, x, x_shape):
        h, w = x_shape
        b, c, hh, ww, p, _ = x.shape
        x = x.permute(0, 1, 2, 4, 3, 5).reshape(b, c, hh * p, ww * p)
        return x[:, :, :h, :w]

    def forward(self, x):
        b, c, h, w = x.shape
        q = _dw(_c1(x, self.q), self.q_dwconv)
        kv = _dw(_c1(x, self.kv), self.kv_dwconv)
        k, v = kv.chunk(2, dim=1)
        q = self.pad_and_rearrange(q)
        k = self.pad_and_rearrange(k)
        out = torch.fft.rfft2(q.float()) * torch.fft.rfft2(k.float())
        out = torch.fft.irfft2(out, s=(self.patch_size, self.patch_size))
        out = self.rearrange_to_original(out, (h, w)).to(x.dtype).contiguous()
        out = self.norm(out)
        out = out * v
        return _c1(out.contiguous(), self.proj_out)


class MySequential(nn.Sequential):
    """nn.Sequential whose layers take (x1, x2) (moce_ir.py:31-50)."""

    def forward(self, x1, x2):
        for layer in self:
            x1 = layer(x1, x2)
        return x1


class ModExpert(nn.Module):
    """Low-rank expert: proj[0] C->r, body, gate by silu(proj[1](shared)), proj[2] r->C, + shortcut (moce_ir.py:520-579)."""

    def __init__(self, dim: int, rank: int, func: nn.Module, depth: int, patch_size: int, kernel_size: int):
        super().__init__()
        self.depth = depth
        self.proj = nn.ModuleList([
            nn.Conv2d(dim, rank, kernel_size=1, padding=0, bias=False),
            nn.Conv2d(dim, rank, kernel_size=1, padding=0, bias=False),
            nn.Conv2d(rank, dim, kernel_size=1, padding=0, bias=False)
        ])
        self.body = func(rank, kernel_size=kernel_size, patch_size=patch_size)

    def process(self, x, shared):
        shortcut = x
        x = _c1(x, self.proj[0])
        x = self.body(x) * F.silu(_c1(shared, self.proj[1]))
        x = _c1(x.contiguous(), self.proj[2])
        return x + shortcut

    def feat_extract(self, feats, shared):
        for _ in range(self.depth):          # the reference re-applies process to the SAME input (moce_ir.py:567-570)
            feat = self.process(feats, shared)
        return feat

    def forward(self, x, shared):
        if x.shape[0] == 0:
            return x
        return self.feat_extract(x.contiguous(), shared.contiguous())


class RoutingFunction(nn.Module):
    """Noisy top-k router (moce_ir.py:684-800).  gate = GAP -> Linear(dim, E); + Linear(freq_dim, E)(freq_emb)."""

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

    def forward(self, x, freq_emb):
        pooled = _GapFn.apply(x.contiguous())                     # native GAP, fp32 [B, C]
        logits = self.gate[2](pooled) + self.freq_gate(freq_emb.float())
        if self.training:
            loss_imp = self.importance_loss(logits.softmax(dim=-1))
        noise = torch.randn_like(logits) * self.noise_std          # train AND eval, as the reference (moce_ir.py:741)
        noisy_logits = logits + noise
        gating_scores = noisy_logits.softmax(dim=-1)
        top_k_values, top_k_indices = torch.topk(gating_scores, self.k, dim=-1)
        if self.training:
            loss_load = self.load_loss(logits, noisy_logits, self.noise_std)
            aux_loss = 0.5 * loss_imp + 0.5 * loss_load
        else:
            aux_loss = 0
        gates = torch.zeros_like(logits).scatter_(1, top_k_indices, top_k_values)
        return gates, top_k_indices, top_k_values, aux_loss

    def importance_loss(self, gating_scores):
        importance = gating_scores.sum(dim=0)
        importance = importance * (self.complexity * self.tau) if self.use_complexity_bias else importance
        return (importance.std() / (importance.mean() + 1e-8)) ** 2

    def load_loss(self, logits, logits_noisy, noise_std):
        thresholds = torch.topk(logits_noisy, self.k, dim=-1).indices[:, -1]
        threshold_per_item = torch.sum(F.one_hot(thresholds, self.num_experts) * logits_noisy, dim=-1)
        noise_required_to_win = (threshold_per_item.unsqueeze(-1) - logits) / noise_std
        p = 1. - Normal(0, 1).cdf(noise_required_to_win)
        p_mean = p.mean(dim=0)
        return (p_mean.std() / (p_mean.mean() + 1e-8)) ** 2


class AdapterLayer(nn.Module):
    """E experts of growing rank/patch/kernel behind the noisy top-k router, then a 1x1 projection (moce_ir.py:584-681)."""

    def __init__(self, dim: int, rank: int, num_experts: int = 4, top_k: int = 2, expert_layer: nn.Module = FFTAttention,
                 stage_depth: int = 1, depth_type: str = "lin", rank_type: str = "constant", freq_dim: int = 128,
                 with_complexity: bool = False, complexity_scale: str = "min"):
        super().__init__()
        self.tau = 1
        self.loss = None
        self.top_k = top_k
        self.noise_eps = 1e-2
        self.num_experts = num_experts
        patch_sizes = [2 ** (i + 2) for i in range(num_experts)]
        kernel_sizes = [3 + (2 * i) for i in range(num_experts)]
        if depth_type == "lin":
            depths = [stage_depth + i for i in range(num_experts)]
        elif depth_type == "double":
            depths = [stage_depth + (2 * i) for i in range(num_experts)]
        elif depth_type == "exp":
            depths = [2 ** (i) for i in range(num_experts)]
        elif depth_type == "fact":
            depths = [math.factorial(i + 1) for i in range(num_experts)]
        elif isinstance(depth_type, int):
            depths = [depth_type for _ in range(num_experts)]
        elif depth_type == "constant":
            depths = [stage_depth for i in range(num_experts)]
        else:
            raise NotImplementedError
        if rank_type == "constant":
            ranks = [rank for _ in range(num_experts)]
        elif rank_type == "lin":
            ranks = [rank + i for i in range(num_experts)]
        elif rank_type == "double":
            ranks = [rank + (2 * i) for i in range(num_experts)]
        elif rank_type == "exp":
            ranks = [rank ** (i + 1) for i in range(num_experts)]
        elif rank_type == "fact":
            ranks = [math.factorial(rank + i) for i in range(num_experts)]
        elif rank_type == "spread":
            ranks = [dim // (2 ** i) for i in range(num_experts)][::-1]
        else:
            raise NotImplementedError
        self.experts = nn.ModuleList([
            MySequential(*[ModExpert(dim, rank=rank, func=expert_layer, depth=depth, patch_size=patch, kernel_size=kernel)])
            for idx, (depth, rank, patch, kernel) in enumerate(zip(depths, ranks, patch_sizes, kernel_sizes))
        ])
        self.proj_out = nn.Conv2d(dim, dim, kernel_size=1, padding=0, bias=False)
        expert_complexity = torch.tensor([sum(p.numel() for p in expert.parameters()) for expert in self.experts])
        self.routing = RoutingFunction(dim, freq_dim, num_experts=num_experts, k=top_k, complexity=expert_complexity,
                                       use_complexity_bias=with_complexity, complexity_scale=complexity_scale)

    def forward(self, x, freq_emb, shared):
        gates, top_k_indices, top_k_values, aux_loss = self.routing(x, freq_emb)
        self.loss = aux_loss
        if self.training:
            dispatcher = SparseDispatcher(self.num_experts, gates)
            expert_inputs = dispatcher.dispatch(x)
            expert_shared_intputs = dispatcher.dispatch(shared)
            expert_outputs = [self.experts[exp](expert_inputs[exp], expert_shared_intputs[exp])
                              for exp in range(len(self.experts))]
            out = dispatcher.combine(expert_outputs, multiply_by_gates=True)
        else:                                   # B == 1 semantics of the reference's test path (moce_ir.py:674-678)
            selected_experts = [self.experts[i] for i in top_k_indices.squeeze(0)]
            expert_outputs = torch.stack([expert(x, shared) for expert in selected_experts], dim=1)
            gates = gates.gather(1, top_k_indices)
            weighted_outputs = gates.unsqueeze(2).unsqueeze(3).unsqueeze(4) * expert_outputs
            out = weighted_outputs.sum(dim=1)
        # combine accumulates in fp32 (moce_ir.py:123); hand the next 1x1 the activation dtype again
        return _c1(out.to(x.dtype).contiguous(), self.proj_out)


class EncoderBlock(nn.Module):
    """x + mixer(norms[0](x)); + ffn(norms[1](.))  (moce_ir.py:805-834): the Restormer block under MoCE's names."""

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norms = nn.ModuleList([LayerNorm(dim, LayerNorm_type), LayerNorm(dim, LayerNorm_type)])
        self.mixer = Attention(dim, num_heads, bias)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x):
        params = self.norms[0]._params() + self.mixer._params() + self.norms[1]._params() + self.ffn._params()
        return _apply(_BlockFn, x, self.mixer.num_heads, *params)


class DecoderBlock(nn.Module):
    """Shared MDTA + MoCE adapter + cross-MDTA mixer + GDFN (moce_ir.py:839-897).  Returns (x, adapter.loss)."""

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
