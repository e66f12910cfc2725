"""Functional CPU restatement of the MoCE-IR / AdaIR block pieces (test oracle; see ``oracle/__init__.py``).

Plain torch on CPU, differentiable, parameters passed as reference-keyed state dicts.  File:line citations are relative
to the upstream repository root (top-level ``moce_ir.py`` unless stated)."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .restormer_ref import gdfn, layernorm_nchw, mdta, mdta_cross, sub_state

Tensor = torch.Tensor

__all__ = ["cross_attention", "fft_attention", "mod_expert", "routing", "dispatch_indices", "adapter_layer", "frequency_embedding",
           "encoder_block", "decoder_block", "expert_ranks"]


def cross_attention(x: Tensor, y: Tensor, sd: Dict[str, Tensor], heads: int) -> Tensor:
    """CrossAttention.forward (moce_ir.py:345-368; AdaIR-main/net/model.py:191-216): keys q, q_dwconv, kv, kv_dwconv,
    project_out, temperature."""
    g = sd.get
    return mdta_cross(x, y, sd["temperature"], sd["q.weight"], sd["q_dwconv.weight"], sd["kv.weight"],
                      sd["kv_dwconv.weight"], sd["project_out.weight"], heads, g("q.bias"), g("q_dwconv.bias"),
                      g("kv.bias"), g("kv_dwconv.bias"), g("project_out.bias"))


def _to_patches(t: Tensor, p: int) -> Tensor:
    b, c, h, w = t.shape
    t = F.pad(t, (0, (p - w % p) % p, 0, (p - h % p) % p))
    hh, ww = t.shape[-2] // p, t.shape[-1] // p
    return t.reshape(b, c, hh, p, ww, p).permute(0, 1, 2, 4, 3, 5)


def fft_attention(x: Tensor, sd: Dict[str, Tensor], patch: int) -> Tensor:
    """FFTAttention.forward (moce_ir.py:402-422): q = dw3(1x1(x)); k,v = dw7(1x1(x)); per p x p patch the product of the
    rfft2 spectra of q and k (a circular convolution), irfft2, un-patch, WithBias LayerNorm, times v, 1x1 out."""
    c = x.shape[1]
    b, _, h, w = x.shape
    q = F.conv2d(F.conv2d(x, sd["q.weight"]), sd["q_dwconv.weight"], sd["q_dwconv.bias"], padding=1, groups=c)
    kv = F.conv2d(F.conv2d(x, sd["kv.weight"]), sd["kv_dwconv.weight"], sd["kv_dwconv.bias"], padding=3, groups=2 * c)
    k, v = kv[:, :c], kv[:, c:]
    qp, kp = _to_patches(q, patch), _to_patches(k, patch)
    out = torch.fft.irfft2(torch.fft.rfft2(qp) * torch.fft.rfft2(kp), s=(patch, patch))
    bb, cc, hh, ww, _, _ = out.shape
    out = out.permute(0, 1, 2, 4, 3, 5).reshape(bb, cc, hh * patch, ww * patch)[:, :, :h, :w]
    out = layernorm_nchw(out, sd["norm.body.weight"], sd["norm.body.bias"], "WithBias") * v
    return F.conv2d(out, sd["proj_out.weight"], sd["proj_out.bias"])


def mod_expert(x: Tensor, shared: Tensor, sd: Dict[str, Tensor], patch: int) -> Tensor:
    """ModExpert.process (moce_ir.py:545-558).  feat_extract repeats it on the same input, so depth does not change the
    result (:567-570)."""
    if x.shape[0] == 0:
        return x
    t = F.conv2d(x, sd["proj.0.weight"])
    t = fft_attention(t, sub_state(sd, "body."), patch) * F.silu(F.conv2d(shared, sd["proj.1.weight"]))
    return F.conv2d(t, sd["proj.2.weight"]) + x


def expert_ranks(dim: int, rank: int, num_experts: int, rank_type: str) -> List[int]:
    """moce_ir.py:631-644 (the schedules the reference configs use)."""
    if rank_type == "constant":
        return [rank] * num_experts
    if rank_type == "spread":
        return [dim // (2 ** i) for i in range(num_experts)][::-1]
    raise NotImplementedError(rank_type)


def routing(x: Tensor, freq_emb: Tensor, sd: Dict[str, Tensor], k: int, noise: Tensor, training: bool,
            complexity: Optional[Tensor] = None, use_complexity_bias: bool = True):
    """RoutingFunction.forward with the N(0,1) draw passed in (moce_ir.py:736-755), plus importance / load losses
    (:759-800).  Returns gates [B,E], top-k indices, top-k values, aux loss."""
    E = sd["gate.2.weight"].shape[0]
    noise_std = 1.0 / E
    logits = x.mean(dim=(2, 3)) @ sd["gate.2.weight"].t() + freq_emb @ sd["freq_gate.weight"].t()
    noisy = logits + noise * noise_std
    scores = noisy.softmax(dim=-1)
    vals, idx = torch.topk(scores, k, dim=-1)
    gates = torch.zeros_like(logits).scatter(1, idx, vals)
    aux = 0
    if training:
        imp = logits.softmax(dim=-1).sum(dim=0)
        if use_complexity_bias:
            imp = imp * complexity
        loss_imp = (imp.std() / (imp.mean() + 1e-8)) ** 2
        thr = torch.topk(noisy, k, dim=-1).indices[:, -1]
        thr_val = noisy.gather(1, thr[:, None])
        z = (thr_val - logits) / noise_std
        p = 1.0 - 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))
        pm = p.mean(dim=0)
        loss_load = (pm.std() / (pm.mean() + 1e-8)) ** 2
        aux = 0.5 * loss_imp + 0.5 * loss_load
    return gates, idx, vals, aux


def dispatch_indices(gates: Tensor) -> List[List[int]]:
    """Per expert, the batch rows with a non-zero gate in increasing order (SparseDispatcher.__init__, moce_ir.py:82-91)."""
    return [[b for b in range(gates.shape[0]) if float(gates[b, e]) > 0] for e in range(gates.shape[1])]


def adapter_layer(x: Tensor, freq_emb: Tensor, shared: Tensor, sd: Dict[str, Tensor], cfg: dict, noise: Tensor,
                  training: bool):
    """AdapterLayer.forward (moce_ir.py:660-681).  cfg: dim, rank, num_experts, top_k, rank_type, with_complexity,
    complexity [E].  Training: every expert runs on its routed rows, outputs are scaled by their gate and scattered back
    in fp32 (:116-124).  Eval: the B == 1 path, the top-k experts of sample 0 applied to the whole batch (:674-678)."""
    E, k = cfg["num_experts"], cfg["top_k"]
    patches = [2 ** (i + 2) for i in range(E)]
    gates, idx, vals, aux = routing(x, freq_emb, sub_state(sd, "routing."), k, noise, training, cfg.get("complexity"),
                                    cfg.get("with_complexity", False))
    if training:
        out = torch.zeros_like(x, dtype=torch.float32 if x.dtype != torch.float64 else torch.float64)
        for e, rows in enumerate(dispatch_indices(gates)):
            if not rows:
                continue
            r = torch.tensor(rows)
            y = mod_expert(x[r], shared[r], sub_state(sd, f"experts.{e}.0."), patches[e])
            out = out.index_add(0, r, (y * gates[r, e].view(-1, 1, 1, 1)).to(out.dtype))
    else:
        sel = [int(i) for i in idx[0]]
        outs = torch.stack([mod_expert(x, shared, sub_state(sd, f"experts.{e}.0."), patches[e]) for e in sel], dim=1)
        out = (gates.gather(1, idx)[:, :, None, None, None] * outs).sum(dim=1)
    return F.conv2d(out.to(x.dtype), sd["proj_out.weight"]), aux


def encoder_block(x: Tensor, sd: Dict[str, Tensor], heads: int, ln_kind: str = "WithBias") -> Tensor:
    """EncoderBlock.forward (moce_ir.py:825-834): keys norms.{0,1}, mixer, ffn."""
    g = sd.get
    y = layernorm_nchw(x, sd["norms.0.body.weight"], g("norms.0.body.bias"), ln_kind)
    x = x + mdta(y, sd["mixer.temperature"], sd["mixer.qkv.weight"], sd["mixer.qkv_dwconv.weight"],
                 sd["mixer.project_out.weight"], heads, g("mixer.qkv.bias"), g("mixer.qkv_dwconv.bias"),
                 g("mixer.project_out.bias"))
    y = layernorm_nchw(x, sd["norms.1.body.weight"], g("norms.1.body.bias"), ln_kind)
    return x + gdfn(y, sd["ffn.project_in.weight"], sd["ffn.dwconv.weight"], sd["ffn.project_out.weight"],
                    g("ffn.project_in.bias"), g("ffn.dwconv.bias"), g("ffn.project_out.bias"))


def decoder_block(x: Tensor, freq_emb: Tensor, sd: Dict[str, Tensor], heads: int, cfg: dict, noise: Tensor,
                  training: bool, ln_kind: str = "WithBias"):
    """DecoderBlock.forward (moce_ir.py:886-897) -> (x, aux loss)."""
    g = sd.get
    shortcut = x
    t = layernorm_nchw(x, sd["norms.0.body.weight"], g("norms.0.body.bias"), ln_kind)
    x_s = F.conv2d(t, sd["proj.0.weight"], sd["proj.0.bias"])
    x_a = F.conv2d(t, sd["proj.1.weight"], sd["proj.1.bias"])
    ss = sub_state(sd, "shared.")
    x_s = mdta(x_s, ss["temperature"], ss["qkv.weight"], ss["qkv_dwconv.weight"], ss["project_out.weight"], heads,
               ss.get("qkv.bias"), ss.get("qkv_dwconv.bias"), ss.get("project_out.bias"))
    x_a, aux = adapter_layer(x_a, freq_emb, x_s, sub_state(sd, "adapter."), cfg, noise, training)
    x = cross_attention(x_a, x_s, sub_state(sd, "mixer."), heads) + shortcut
    t = layernorm_nchw(x, sd["norms.1.body.weight"], g("norms.1.body.bias"), ln_kind)
    x = x + gdfn(t, sd["ffn.project_in.weight"], sd["ffn.dwconv.weight"], sd["ffn.project_out.weight"],
                 g("ffn.project_in.bias"), g("ffn.dwconv.bias"), g("ffn.project_out.bias"))
    return x, aux


def frequency_embedding(x: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """FrequencyEmbedding.forward (moce_ir.py:1070-1075): depthwise 3x3 high-pass -> GELU -> mean over the plane -> MLP."""
    c = x.shape[1]
    h = F.gelu(F.conv2d(x, sd["high_conv.0.conv.weight"], None, padding=1, groups=c)).mean(dim=(-2, -1))
    h = F.gelu(F.linear(h, sd["mlp.0.weight"], sd["mlp.0.bias"]))
    return F.linear(h, sd["mlp.2.weight"], sd["mlp.2.bias"])


def expert_complexity(sd: Dict[str, Tensor], prefix: str, num_experts: int, scale: str = "max") -> Tensor:
    """AdapterLayer's `complexity` buffer (moce_ir.py:653, RoutingFunction.__init__ :713-724): the experts' parameter counts,
    normalised by the largest ("max") or smallest ("min").  Computed in fp32 like the reference's buffer."""
    counts = [sum(v.numel() for k, v in sd.items() if k.startswith(f"{prefix}experts.{e}.")) for e in range(num_experts)]
    c = torch.tensor(counts, dtype=torch.float32)
    return c / (c.max() if scale == "max" else c.min())


def moceir_forward(img: Tensor, sd: Dict[str, Tensor], cfg: dict, noise: Tensor, training: bool):
    """MoCEIR.forward (moce_ir.py:1207-1231) -> (restored, total_loss): patch embedding (3x3, :1209), encoder groups of
    EncoderBlocks + Downsample (3x3 C -> C/2, PixelUnshuffle(2), :1013-1028) (:1213-1216), latent group, FrequencyEmbedding of
    the latent features (:1218-1219), per decoder stage Upsample (3x3 C -> 2C, PixelShuffle(2), :1031-1042) -> cat with the
    skip -> 1x1 fusion -> DecoderBlocks whose auxiliary losses are summed (:1221-1225), refinement group, 3x3 output conv +
    input (:1227-1228), total_loss / sum(num_dec_blocks) (:1230).
    cfg: dim, levels, heads, num_blocks, num_dec_blocks, num_refinement_blocks, rank, num_experts, rank_type, topk,
    with_complexity, complexity_scale.  `noise` [B, E] is the N(0,1) draw every router adds (the tests inject one seeded draw
    for all decoder blocks, exactly as the golden capture patched torch.randn_like)."""
    levels, heads = cfg["levels"], list(cfg["heads"])
    nb, ndb = list(cfg["num_blocks"]), list(cfg["num_dec_blocks"])
    E = cfg["num_experts"]
    dt = img.dtype

    def group(x, prefix, n, hd):
        for j in range(n):
            x = encoder_block(x, sub_state(sd, f"{prefix}layers.{j}."), hd)
        return x

    feats = F.conv2d(img, sd["patch_embed.proj.weight"], None, padding=1)
    skips = []
    for i in range(levels - 1):
        feats = group(feats, f"enc.{i}.0.", nb[i], heads[i])
        skips.append(feats)
        feats = F.pixel_unshuffle(F.conv2d(feats, sd[f"enc.{i}.1.body.0.weight"], None, padding=1), 2)
    feats = group(feats, "latent.", nb[-1], heads[-1])
    freq_emb = frequency_embedding(feats, sub_state(sd, "freq_embed."))
    total = 0
    rheads, rndb = heads[::-1], ndb[::-1]
    for j in range(levels - 1):
        dim = cfg["dim"] * 2 ** (levels - 2 - j)
        feats = F.pixel_shuffle(F.conv2d(feats, sd[f"dec.{j}.0.body.0.weight"], None, padding=1), 2)
        feats = F.conv2d(torch.cat([feats, skips.pop()], dim=1), sd[f"dec.{j}.1.weight"], sd.get(f"dec.{j}.1.bias"))
        for b in range(rndb[j]):
            pre = f"dec.{j}.2.layers.{b}."
            bsd = sub_state(sd, pre)
            acfg = dict(dim=dim, rank=cfg["rank"], num_experts=E, top_k=cfg["topk"], rank_type=cfg["rank_type"],
                        with_complexity=cfg.get("with_complexity", False),
                        complexity=expert_complexity(bsd, "adapter.", E, cfg.get("complexity_scale", "max")).to(dt))
            feats, aux = decoder_block(feats, freq_emb, bsd, rheads[j + 1], acfg, noise, training)
            total = total + aux
    feats = group(feats, "refinement.", cfg["num_refinement_blocks"], heads[0])
    out = F.conv2d(feats, sd["output.weight"], sd.get("output.bias"), padding=1) + img
    return out, total / sum(ndb)
