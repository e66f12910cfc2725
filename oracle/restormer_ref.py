"""Functional CPU restatement of the Restormer block path (test oracle).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Plain torch ops on CPU
tensors, written from the math of the reference (file:line cited per function,
relative to the upstream repo root).  Parameters are passed as a flat
``state_dict`` that uses the reference's own key names, so the same dict can
be loaded into the reference module, the oracle and the HIP-backed modules.

All functions are differentiable through torch autograd, which is how the
backward oracle is obtained (fp64 capable).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

__all__ = [
    "layernorm_nchw", "gdfn", "mdta", "mdta_cross", "transformer_block",
    "restormer_forward", "restormer_config", "RESTORMER_BASE", "RESTORMER_TINY",
    "make_restormer_state", "psnr", "ssim", "degrade_sigma", "sub_state",
]

LN_EPS = 1e-5          # Restormer.py:39,57  (eps sits INSIDE the sqrt)
NORMALIZE_EPS = 1e-12  # torch.nn.functional.normalize default, Restormer.py:121-122


# --------------------------------------------------------------------------
# LayerNorm over the channel dim of an NCHW map
# --------------------------------------------------------------------------
def layernorm_nchw(x: Tensor, weight: Tensor, bias: Optional[Tensor], kind: str = "WithBias") -> Tensor:
    """Per-pixel LayerNorm over C (Restormer.py:37-39 BiasFree, :54-57 WithBias, :68-70).

    The reference moves C last (to_3d), normalises, moves it back; here the
    reduction is taken directly over dim 1.  Variance is the biased one about
    the mean in both flavours; BiasFree does NOT centre x (Restormer.py:38-39).
    """
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    w = weight.view(1, -1, 1, 1)
    if kind == "BiasFree":
        return x / torch.sqrt(var + LN_EPS) * w
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + bias.view(1, -1, 1, 1)


# --------------------------------------------------------------------------
# GDFN
# --------------------------------------------------------------------------
def gdfn(x: Tensor, w_in: Tensor, w_dw: Tensor, w_out: Tensor,
         b_in: Optional[Tensor] = None, b_dw: Optional[Tensor] = None,
         b_out: Optional[Tensor] = None) -> Tensor:
    """Gated-Dconv feed-forward (Restormer.py:76-93).

    project_in 1x1 C->2h, depthwise 3x3 (pad 1) on 2h, split in halves
    (first h channels gate through exact-erf GELU, :90-91), project_out 1x1 h->C.
    """
    two_h = w_in.shape[0]
    t = F.conv2d(x, w_in, b_in)
    t = F.conv2d(t, w_dw, b_dw, padding=w_dw.shape[-1] // 2, groups=two_h)
    x1, x2 = t[:, : two_h // 2], t[:, two_h // 2:]
    g = 0.5 * x1 * (1.0 + torch.erf(x1 / math.sqrt(2.0))) * x2
    return F.conv2d(g, w_out, b_out)


# --------------------------------------------------------------------------
# MDTA
# --------------------------------------------------------------------------
def _l2_normalize_rows(t: Tensor) -> Tensor:
    # F.normalize(dim=-1): x / max(||x||_2, eps)   (Restormer.py:121-122)
    # (Tensor.norm, not sum-of-squares + sqrt: its backward is 0 at an all-zero row - AdaIR's empty low band - where sqrt'(0) = inf)
    n = t.norm(p=2, dim=-1, keepdim=True).clamp_min(NORMALIZE_EPS)
    return t / n


def _channel_attention(q: Tensor, k: Tensor, v: Tensor, temperature: Tensor, heads: int) -> Tensor:
    """q,k,v: [B,C,H,W] -> [B,C,H,W].  Head-major channel split, rows L2-normalised
    over N=H*W, C/heads x C/heads attention scaled by temperature[head]
    (Restormer.py:117-129)."""
    b, c, hh, ww = q.shape
    cph = c // heads
    q = _l2_normalize_rows(q.reshape(b, heads, cph, hh * ww))
    k = _l2_normalize_rows(k.reshape(b, heads, cph, hh * ww))
    v = v.reshape(b, heads, cph, hh * ww)
    attn = torch.matmul(q, k.transpose(-1, -2)) * temperature.view(1, heads, 1, 1)
    attn = torch.softmax(attn, dim=-1)
    return torch.matmul(attn, v).reshape(b, c, hh, ww)


def mdta(x: Tensor, temperature: Tensor, w_qkv: Tensor, w_dw: Tensor, w_out: Tensor, heads: int,
         b_qkv: Optional[Tensor] = None, b_dw: Optional[Tensor] = None,
         b_out: Optional[Tensor] = None) -> Tensor:
    """Multi-Dconv-head transposed self-attention (Restormer.py:99-132)."""
    c3 = w_qkv.shape[0]
    t = F.conv2d(x, w_qkv, b_qkv)
    t = F.conv2d(t, w_dw, b_dw, padding=w_dw.shape[-1] // 2, groups=c3)
    c = c3 // 3
    out = _channel_attention(t[:, :c], t[:, c:2 * c], t[:, 2 * c:], temperature, heads)
    return F.conv2d(out, w_out, b_out)


def mdta_cross(x: Tensor, y: Tensor, temperature: Tensor, w_q: Tensor, w_q_dw: Tensor,
               w_kv: Tensor, w_kv_dw: Tensor, w_out: Tensor, heads: int,
               b_q=None, b_q_dw=None, b_kv=None, b_kv_dw=None, b_out=None) -> Tensor:
    """Cross MDTA: q from x, k/v from y (moce_ir.py:325-368 with a 7x7 kv
    depthwise conv; AdaIR-main/net/model.py:177-216 with 3x3)."""
    c = w_q.shape[0]
    q = F.conv2d(x, w_q, b_q)
    q = F.conv2d(q, w_q_dw, b_q_dw, padding=w_q_dw.shape[-1] // 2, groups=c)
    kv = F.conv2d(y, w_kv, b_kv)
    kv = F.conv2d(kv, w_kv_dw, b_kv_dw, padding=w_kv_dw.shape[-1] // 2, groups=2 * c)
    out = _channel_attention(q, kv[:, :c], kv[:, c:], temperature, heads)
    return F.conv2d(out, w_out, b_out)


# --------------------------------------------------------------------------
# TransformerBlock and the Restormer U-Net, driven by a reference-keyed state dict
# --------------------------------------------------------------------------
def sub_state(sd: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    """Entries of ``sd`` under ``prefix`` with the prefix stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def transformer_block(x: Tensor, sd: Dict[str, Tensor], heads: int, ln_kind: str = "WithBias") -> Tensor:
    """x + attn(norm1(x)); then x + ffn(norm2(x))  (Restormer.py:146-150).

    ``sd`` holds the block's parameters under the reference names
    norm1.body.*, attn.*, norm2.body.*, ffn.* (SURVEY 8(b))."""
    g = sd.get
    y = layernorm_nchw(x, sd["norm1.body.weight"], g("norm1.body.bias"), ln_kind)
    x = x + mdta(y, sd["attn.temperature"], sd["attn.qkv.weight"], sd["attn.qkv_dwconv.weight"],
                 sd["attn.project_out.weight"], heads, g("attn.qkv.bias"), g("attn.qkv_dwconv.bias"),
                 g("attn.project_out.bias"))
    y = layernorm_nchw(x, sd["norm2.body.weight"], g("norm2.body.bias"), ln_kind)
    x = x + gdfn(y, sd["ffn.project_in.weight"], sd["ffn.dwconv.weight"], sd["ffn.project_out.weight"],
                 g("ffn.project_in.bias"), g("ffn.dwconv.bias"), g("ffn.project_out.bias"))
    return x


def restormer_config(dim=48, num_blocks=(4, 6, 6, 8), num_refinement_blocks=4, heads=(1, 2, 4, 8),
                     ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias",
                     inp_channels=3, out_channels=3):
    """Constructor arguments of the reference network (Restormer.py:194-205)."""
    return dict(dim=dim, num_blocks=list(num_blocks), num_refinement_blocks=num_refinement_blocks,
                heads=list(heads), ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                LayerNorm_type=LayerNorm_type, inp_channels=inp_channels, out_channels=out_channels)


RESTORMER_BASE = restormer_config()
# "Restormer-tiny" is pinned by SURVEY section 8: dim 16, 2 blocks per level, 2 refinement blocks.
RESTORMER_TINY = restormer_config(dim=16, num_blocks=(2, 2, 2, 2), num_refinement_blocks=2)


def _stage(x: Tensor, sd: Dict[str, Tensor], prefix: str, n: int, heads: int, ln_kind: str) -> Tensor:
    for i in range(n):
        x = transformer_block(x, sub_state(sd, f"{prefix}.{i}."), heads, ln_kind)
    return x


def restormer_forward(img: Tensor, sd: Dict[str, Tensor], cfg: dict) -> Tensor:
    """Whole-network forward (Restormer.py:245-284), non dual-pixel branch.

    patch_embed 3x3 -> 3 encoder levels (3x3 conv C->C/2 + PixelUnshuffle between,
    :171-179) -> latent -> 3 decoder levels (3x3 conv C->2C + PixelShuffle, :181-189;
    skip concat; 1x1 channel reduce at levels 3 and 2 only, :223,228,231) ->
    refinement -> 3x3 output conv + input residual (:281)."""
    nb, hd, ln = cfg["num_blocks"], cfg["heads"], cfg["LayerNorm_type"]
    g = sd.get

    def conv3(t, key):
        return F.conv2d(t, sd[key + ".weight"], g(key + ".bias"), padding=1)

    e1 = _stage(conv3(img, "patch_embed.proj"), sd, "encoder_level1", nb[0], hd[0], ln)
    e2 = _stage(F.pixel_unshuffle(conv3(e1, "down1_2.body.0"), 2), sd, "encoder_level2", nb[1], hd[1], ln)
    e3 = _stage(F.pixel_unshuffle(conv3(e2, "down2_3.body.0"), 2), sd, "encoder_level3", nb[2], hd[2], ln)
    lat = _stage(F.pixel_unshuffle(conv3(e3, "down3_4.body.0"), 2), sd, "latent", nb[3], hd[3], ln)

    d3 = torch.cat([F.pixel_shuffle(conv3(lat, "up4_3.body.0"), 2), e3], 1)
    d3 = F.conv2d(d3, sd["reduce_chan_level3.weight"], g("reduce_chan_level3.bias"))
    d3 = _stage(d3, sd, "decoder_level3", nb[2], hd[2], ln)

    d2 = torch.cat([F.pixel_shuffle(conv3(d3, "up3_2.body.0"), 2), e2], 1)
    d2 = F.conv2d(d2, sd["reduce_chan_level2.weight"], g("reduce_chan_level2.bias"))
    d2 = _stage(d2, sd, "decoder_level2", nb[1], hd[1], ln)

    d1 = torch.cat([F.pixel_shuffle(conv3(d2, "up2_1.body.0"), 2), e1], 1)
    d1 = _stage(d1, sd, "decoder_level1", nb[0], hd[0], ln)
    d1 = _stage(d1, sd, "refinement", cfg["num_refinement_blocks"], hd[0], ln)
    return conv3(d1, "output") + img


# --------------------------------------------------------------------------
# Seeded parameters, synthetic data, PSNR
# --------------------------------------------------------------------------
def _block_shapes(c: int, heads: int, ffn: float, bias: bool, ln_kind: str):
    """Parameter names/shapes of one block in the reference's registration order
    (norm1, attn, norm2, ffn: Restormer.py:141-144; conv bias follows its weight)."""
    h = int(c * ffn)  # Restormer.py:80
    shapes: Dict[str, tuple] = {}

    def ln(name):
        shapes[name + ".body.weight"] = (c,)
        if ln_kind != "BiasFree":
            shapes[name + ".body.bias"] = (c,)

    def conv(name, shape):
        shapes[name + ".weight"] = shape
        if bias:
            shapes[name + ".bias"] = (shape[0],)

    ln("norm1")
    shapes["attn.temperature"] = (heads, 1, 1)
    conv("attn.qkv", (3 * c, c, 1, 1))
    conv("attn.qkv_dwconv", (3 * c, 1, 3, 3))
    conv("attn.project_out", (c, c, 1, 1))
    ln("norm2")
    conv("ffn.project_in", (2 * h, c, 1, 1))
    conv("ffn.dwconv", (2 * h, 1, 3, 3))
    conv("ffn.project_out", (c, h, 1, 1))
    return shapes


def restormer_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """state_dict key -> shape, in the reference's registration order (Restormer.py:209-243)."""
    d, nb, hd = cfg["dim"], cfg["num_blocks"], cfg["heads"]
    ffn, bias, ln = cfg["ffn_expansion_factor"], cfg["bias"], cfg["LayerNorm_type"]
    out: Dict[str, tuple] = {}

    def blocks(prefix, n, c, heads):
        for i in range(n):
            for k, s in _block_shapes(c, heads, ffn, bias, ln).items():
                out[f"{prefix}.{i}.{k}"] = s

    def conv(key, co, ci, k, b=False):
        out[key + ".weight"] = (co, ci, k, k)
        if b:
            out[key + ".bias"] = (co,)

    conv("patch_embed.proj", d, cfg["inp_channels"], 3)
    blocks("encoder_level1", nb[0], d, hd[0])
    conv("down1_2.body.0", d // 2, d, 3)
    blocks("encoder_level2", nb[1], 2 * d, hd[1])
    conv("down2_3.body.0", d, 2 * d, 3)
    blocks("encoder_level3", nb[2], 4 * d, hd[2])
    conv("down3_4.body.0", 2 * d, 4 * d, 3)
    blocks("latent", nb[3], 8 * d, hd[3])
    conv("up4_3.body.0", 16 * d, 8 * d, 3)
    conv("reduce_chan_level3", 4 * d, 8 * d, 1, bias)
    blocks("decoder_level3", nb[2], 4 * d, hd[2])
    conv("up3_2.body.0", 8 * d, 4 * d, 3)
    conv("reduce_chan_level2", 2 * d, 4 * d, 1, bias)
    blocks("decoder_level2", nb[1], 2 * d, hd[1])
    conv("up2_1.body.0", 4 * d, 2 * d, 3)
    blocks("decoder_level1", nb[0], 2 * d, hd[0])
    blocks("refinement", cfg["num_refinement_blocks"], 2 * d, hd[0])
    conv("output", cfg["out_channels"], 2 * d, 3, bias)
    return out


def seeded_tensor(rng, shape, kind: str, dtype=torch.float32) -> Tensor:
    """Version-stable parameter fill from ``numpy.random.default_rng``.

    kind: 'conv' -> N(0, 1/fan_in) ; 'ln_w' -> 1 + 0.2 N ; 'ln_b'/'bias' -> 0.1 N ;
    'temp' -> U[0.5, 2]  (non-trivial temperatures)."""
    import numpy as np
    if kind == "conv":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        a = rng.standard_normal(shape) / math.sqrt(fan_in)
    elif kind == "ln_w":
        a = 1.0 + 0.2 * rng.standard_normal(shape)
    elif kind == "temp":
        a = rng.uniform(0.5, 2.0, shape)
    else:
        a = 0.1 * rng.standard_normal(shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype)


def _kind_of(key: str) -> str:
    if key.endswith("temperature"):
        return "temp"
    if ".body.weight" in key and "norm" in key:
        return "ln_w"
    if ".body.bias" in key and "norm" in key:
        return "ln_b"
    if key.endswith(".bias"):
        return "bias"
    return "conv"


def make_state(shapes: Dict[str, tuple], seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    import numpy as np
    rng = np.random.default_rng(seed)
    return {k: seeded_tensor(rng, s, _kind_of(k), dtype) for k, s in shapes.items()}


def make_restormer_state(cfg: dict, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    return make_state(restormer_param_shapes(cfg), seed, dtype)


def make_block_state(c: int, heads: int, ffn: float = 2.66, bias: bool = False, ln_kind: str = "WithBias",
                     seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    return make_state(_block_shapes(c, heads, ffn, bias, ln_kind), seed, dtype)


def degrade_sigma(clean: Tensor, sigma: float, seed: int) -> Tensor:
    """Gaussian-noise degradation on the uint8 grid:
    clip(round(255 x) + sigma N(0,1), 0, 255) -> uint8 -> /255
    (MoCE-IR-main/src/data/degradation_utils.py:21-24: noise added to a uint8 HWC patch,
    clipped, cast back to uint8; ToTensor then divides by 255)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    x = np.round(clean.detach().cpu().double().numpy() * 255.0)
    noisy = np.clip(x + sigma * rng.standard_normal(x.shape), 0, 255).astype(np.uint8)
    return torch.from_numpy(noisy.astype(np.float32) / 255.0).to(clean.dtype)


def psnr(restored: Tensor, clean: Tensor) -> float:
    """10 log10(1/MSE) on outputs clamped to [0,1], data_range 1
    (AdaIR-main/utils/val_utils.py:50-64)."""
    r = restored.detach().double().clamp(0, 1)
    c = clean.detach().double().clamp(0, 1)
    mse = ((r - c) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


def ssim(restored: Tensor, clean: Tensor) -> float:
    """Mean SSIM over a batch as scikit-image's ``structural_similarity(clean, restored, data_range=1, multichannel=True)``
    computes it for float images (the metric AdaIR-main/utils/val_utils.py:50-64 calls; scikit-image is a third-party
    dependency that is NOT installed here - pinned in the upstream requirements as scikit-image 0.19-0.21 - so this is a
    restatement of its published algorithm, Wang et al. 2004 with the library's defaults, and is PARITY UNPINNED):
    7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance (N / (N - 1)), SSIM map averaged over the pixels whose window
    lies inside the image (3-pixel border dropped) and over the channels; inputs clipped to [0, 1] first."""
    x = restored.detach().double().clamp(0, 1)
    y = clean.detach().double().clamp(0, 1)
    k = torch.ones(1, 1, 7, 7, dtype=torch.float64) / 49.0
    B, C, H, W = x.shape

    def box(t):
        return F.conv2d(t.reshape(B * C, 1, H, W), k)           # 'valid' = the cropped interior
    ux, uy = box(x), box(y)
    cn = 49.0 / 48.0
    vx = cn * (box(x * x) - ux * ux)
    vy = cn * (box(y * y) - uy * uy)
    vxy = cn * (box(x * y) - ux * uy)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())
