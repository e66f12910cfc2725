"""CPU restatement of AdaIR's frequency modules and network assembly (TEST INFRASTRUCTURE - only tests/, smoke() and bench's
cpu_baseline may import it).  Functional, plain torch ops, driven by reference-keyed state dicts; every function cites the
reference lines it follows (AdaIR-main/net/model.py).  Pinned by tests/golden/adair_* (captured from the imported reference by
tools/capture_golden_adair.py)."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from . import restormer_ref as R


def _sub(sd: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return R.sub_state(sd, prefix)


def cross(x: Tensor, y: Tensor, sd: Dict[str, Tensor], heads: int) -> Tensor:
    """Chanel_Cross_Attention.forward (model.py:192-216): q from x, k / v from y, 3x3 depthwise on both."""
    g = sd.get
    return R.mdta_cross(x, y, sd["temperature"], sd["q.weight"], sd["q_dwconv.weight"], sd["kv.weight"], sd["kv_dwconv.weight"],
                        sd["project_out.weight"], heads, g("q.bias"), g("q_dwconv.bias"), g("kv.bias"), g("kv_dwconv.bias"),
                        g("project_out.bias"))


def spatial_gate(x: Tensor, w: Tensor) -> Tensor:
    """SpatialGate.forward (model.py:239-245): sigmoid(conv7x7([max_c x, mean_c x]))."""
    s = torch.cat((x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)), 1)
    return torch.sigmoid(F.conv2d(s, w, None, padding=3))


def channel_gate(x: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    """ChannelGate.forward (model.py:262-268): sigmoid(mlp(avgpool x) + mlp(maxpool x)), mlp = 1x1 -> ReLU -> 1x1, no biases."""
    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, w1)), w2)
    return torch.sigmoid(mlp(x.mean((2, 3), keepdim=True)) + mlp(x.amax((2, 3), keepdim=True)))


def fre_refine(low: Tensor, high: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """FreRefine.forward (model.py:282-290)."""
    sw = spatial_gate(high, sd["SpatialGate.spatial.weight"])
    cw = channel_gate(low, sd["ChannelGate.mlp.0.weight"], sd["ChannelGate.mlp.2.weight"])
    return F.conv2d(low * sw + high * cw, sd["proj.weight"], sd.get("proj.bias"))


def mask_half_sizes(feat: Tensor, sd: Dict[str, Tensor], n: int = 128) -> Tensor:
    """Half sizes (h_, w_) of the centred low-frequency rectangle of every sample, int64 [B, 2] (model.py:346-353):
    threshold = sigmoid(rate_conv(avgpool(feat))); h_ = int(h // n * t0), w_ = int(w // n * t1)."""
    h, w = feat.shape[-2:]
    t = F.adaptive_avg_pool2d(feat, 1)
    t = torch.sigmoid(F.conv2d(F.gelu(F.conv2d(t, sd["rate_conv.0.weight"])), sd["rate_conv.2.weight"]))
    return torch.stack(((h // n * t[:, 0, 0, 0]).int(), (w // n * t[:, 1, 0, 0]).int()), 1).long()


def fre_split(feat: Tensor, half: Tensor):
    """(high, low) of FreModule.fft (model.py:355-372): fft2 (norm='forward'), shift to the centre, keep / drop the rectangle
    [h/2 - h_, h/2 + h_) x [w/2 - w_, w/2 + w_), unshift, ifft2 (norm='forward'), magnitude."""
    B, C, h, w = feat.shape
    mask = torch.zeros_like(feat)
    for i in range(B):
        h_, w_ = int(half[i, 0]), int(half[i, 1])
        mask[i, :, h // 2 - h_:h // 2 + h_, w // 2 - w_:w // 2 + w_] = 1
    spec = torch.roll(torch.fft.fft2(feat, norm="forward", dim=(-2, -1)), shifts=(h // 2, w // 2), dims=(2, 3))
    def back(s):
        return torch.abs(torch.fft.ifft2(torch.roll(s, shifts=(-(h // 2), -(w // 2)), dims=(2, 3)), norm="forward", dim=(-2, -1)))
    return back(spec * (1 - mask)), back(spec * mask)


def fre_module(img: Tensor, y: Tensor, sd: Dict[str, Tensor], heads: int) -> Tensor:
    """FreModule.forward (model.py:319-331).  (`conv` and `score_gen` are registered but unused by the reference's forward.)"""
    H, W = y.shape[-2:]
    x = F.interpolate(img, (H, W), mode="bilinear")
    feat = F.conv2d(x, sd["conv1.weight"], None, padding=1)
    high, low = fre_split(feat, mask_half_sizes(feat, sd))
    high = cross(high, y, _sub(sd, "channel_cross_l."), heads)
    low = cross(low, y, _sub(sd, "channel_cross_h."), heads)
    agg = fre_refine(low, high, _sub(sd, "frequency_refine."))
    out = cross(y, agg, _sub(sd, "channel_cross_agg."), heads)
    return out * sd["para1"] + y * sd["para2"]


def adair_forward(img: Tensor, sd: Dict[str, Tensor], cfg: dict) -> Tensor:
    """AdaIR.forward (model.py:448-496): the Restormer U-Net with a FreModule after the latent stage and after decoder levels 3
    and 2 (all three use heads[2], model.py:402-404)."""
    nb, hd, ln = cfg["num_blocks"], cfg["heads"], cfg["LayerNorm_type"]
    g = sd.get

    def conv3(t, key):
        return F.conv2d(t, sd[key + ".weight"], g(key + ".bias"), padding=1)

    e1 = R._stage(conv3(img, "patch_embed.proj"), sd, "encoder_level1", nb[0], hd[0], ln)
    e2 = R._stage(F.pixel_unshuffle(conv3(e1, "down1_2.body.0"), 2), sd, "encoder_level2", nb[1], hd[1], ln)
    e3 = R._stage(F.pixel_unshuffle(conv3(e2, "down2_3.body.0"), 2), sd, "encoder_level3", nb[2], hd[2], ln)
    lat = R._stage(F.pixel_unshuffle(conv3(e3, "down3_4.body.0"), 2), sd, "latent", nb[3], hd[3], ln)
    if cfg.get("decoder", True):
        lat = fre_module(img, lat, _sub(sd, "fre1."), hd[2])
    d3 = torch.cat([F.pixel_shuffle(conv3(lat, "up4_3.body.0"), 2), e3], 1)
    d3 = F.conv2d(d3, sd["reduce_chan_level3.weight"], g("reduce_chan_level3.bias"))
    d3 = R._stage(d3, sd, "decoder_level3", nb[2], hd[2], ln)
    if cfg.get("decoder", True):
        d3 = fre_module(img, d3, _sub(sd, "fre2."), hd[2])
    d2 = torch.cat([F.pixel_shuffle(conv3(d3, "up3_2.body.0"), 2), e2], 1)
    d2 = F.conv2d(d2, sd["reduce_chan_level2.weight"], g("reduce_chan_level2.bias"))
    d2 = R._stage(d2, sd, "decoder_level2", nb[1], hd[1], ln)
    if cfg.get("decoder", True):
        d2 = fre_module(img, d2, _sub(sd, "fre3."), hd[2])
    d1 = torch.cat([F.pixel_shuffle(conv3(d2, "up2_1.body.0"), 2), e1], 1)
    d1 = R._stage(d1, sd, "decoder_level1", nb[0], hd[0], ln)
    d1 = R._stage(d1, sd, "refinement", cfg["num_refinement_blocks"], hd[0], ln)
    return conv3(d1, "output") + img


# --------------------------------------------------------------------------
# Parameter names / shapes in the reference's registration order (model.py:295-317, 378-446)
# --------------------------------------------------------------------------
def fre_param_shapes(dim: int, heads: int, bias: bool = False, in_dim: int = 3) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {"para1": (dim, 1, 1), "para2": (dim, 1, 1), "conv.weight": (dim, in_dim, 3, 3),
                           "conv1.weight": (dim, in_dim, 3, 3), "score_gen.weight": (2, 2, 7, 7), "score_gen.bias": (2,)}
    for name in ("channel_cross_l", "channel_cross_h", "channel_cross_agg"):
        for k, shp in (("temperature", (heads, 1, 1)), ("q", (dim, dim, 1, 1)), ("q_dwconv", (dim, 1, 3, 3)),
                       ("kv", (2 * dim, dim, 1, 1)), ("kv_dwconv", (2 * dim, 1, 3, 3)), ("project_out", (dim, dim, 1, 1))):
            if k == "temperature":
                s[f"{name}.temperature"] = shp
            else:
                s[f"{name}.{k}.weight"] = shp
                if bias:
                    s[f"{name}.{k}.bias"] = (shp[0],)
    s["frequency_refine.SpatialGate.spatial.weight"] = (1, 2, 7, 7)
    s["frequency_refine.ChannelGate.mlp.0.weight"] = (dim // 16, dim, 1, 1)
    s["frequency_refine.ChannelGate.mlp.2.weight"] = (dim, dim // 16, 1, 1)
    s["frequency_refine.proj.weight"] = (dim, dim, 1, 1)
    s["frequency_refine.proj.bias"] = (dim,)
    s["rate_conv.0.weight"] = (dim // 8, dim, 1, 1)
    s["rate_conv.2.weight"] = (2, dim // 8, 1, 1)
    return s


def adair_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """The Restormer parameters with fre1 / fre2 / fre3 registered right after the patch embedding (model.py:399-404)."""
    base = R.restormer_param_shapes({k: v for k, v in cfg.items() if k != "decoder"} | {"inp_channels": 3, "out_channels": 3})
    out: Dict[str, tuple] = {}
    d, hd, bias = cfg["dim"], cfg["heads"], cfg["bias"]
    for k, v in base.items():
        out[k] = v
        if k == "patch_embed.proj.weight" and cfg.get("decoder", True):
            for name, mult in (("fre1", 8), ("fre2", 4), ("fre3", 2)):
                for kk, vv in fre_param_shapes(d * mult, hd[2], bias).items():
                    out[f"{name}.{kk}"] = vv
    return out
