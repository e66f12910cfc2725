"""Drop-in replacements for the classes of the reference's ``AdaIR-main/net/model.py``.

``TransformerBlock`` / ``LayerNorm`` / ``FeedForward`` / ``Attention`` are the Restormer blocks (identical interface,
model.py:25-172).  ``Chanel_Cross_Attention`` (model.py:177-216) is MDTA with q from the image features ``x`` and k,v
from the frequency features ``y`` (3x3 depthwise on both branches).  ``SpatialGate`` / ``ChannelGate`` / ``FreRefine`` /
``FreModule`` (model.py:230-372) and the ``AdaIR`` network (model.py:378-496) keep the reference's constructors, parameter
names and ``forward`` signatures; their plane-sized work runs on ``csrc/adair.hip`` plus the convolution, depthwise and
cross-attention kernels of the other files:

  * the frequency split needs no FFT: the reference's mask keeps at most (2 h/128) x (2 w/128) centred frequencies - none
    at all on feature maps under 128 pixels, i.e. in every training step - so the low band is a direct DFT at those few
    frequencies and ``high = |x - L|``, ``low = |L|`` (``mi_fre_split_*``); the rectangle's size stays on the device;
  * ``F.interpolate(img, (H, W), 'bilinear')`` at the integer factors of the U-Net levels is a 2 x 2 box (``mi_box_down``);
  * SpatialGate's dense 7x7 conv over the [max, mean] planes is the native depthwise 7x7 followed by a channel sum that
    lives inside FreRefine's mixing kernel."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .restormer import (Attention, Downsample, FeedForward, LayerNorm, OverlapPatchEmbed, TransformerBlock, Upsample,  # noqa: F401
                        _apply, _Conv1x1Fn, _conv2d, _CrossAttentionFn, _DwConvFn, _grad_mode, _main_grads, _stage, _up_cat)

__all__ = ["Attention", "FeedForward", "LayerNorm", "TransformerBlock", "Chanel_Cross_Attention", "SpatialGate", "ChannelGate",
           "FreRefine", "FreModule", "AdaIR"]


class Chanel_Cross_Attention(nn.Module):
    def __init__(self, dim, num_head, bias):
        super().__init__()
        self.num_head = num_head
        self.temperature = nn.Parameter(torch.ones(num_head, 1, 1), requires_grad=True)
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=bias)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=3, stride=1, padding=1, groups=dim * 2, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def _params(self):
        return (self.temperature, self.q.weight, self.q.bias, self.q_dwconv.weight, self.q_dwconv.bias, self.kv.weight,
                self.kv.bias, self.kv_dwconv.weight, self.kv_dwconv.bias, self.project_out.weight, self.project_out.bias)

    def forward(self, x, y):
        # x -> q, y -> kv
        assert x.shape == y.shape, 'The shape of feature maps from image and features are not equal!'
        return _apply(_CrossAttentionFn, x, y, self.num_head, *self._params())


# ----------------------------------------------------------------------------------------------- autograd nodes
class _FreSplitFn(torch.autograd.Function):
    """(high, low) of FreModule.fft (model.py:343-372); ``half`` = int32 [B, 2] rectangle half sizes or None (empty)."""

    @staticmethod
    def forward(ctx, feat, half):
        high, low, coef = ops.fre_split_fwd(feat, half)
        if _grad_mode() and ctx.needs_input_grad[0]:
            ctx.save_for_backward(feat, half, coef)
        return high, low

    @staticmethod
    def backward(ctx, dhigh, dlow):
        feat, half, coef = ctx.saved_tensors
        return ops.fre_split_bwd(feat, half, coef, dhigh.contiguous(), dlow.contiguous()), None


class _MaxMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out, idx = ops.chan_maxmean_fwd(x)
        ctx.save_for_backward(idx)
        ctx.c = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return ops.chan_maxmean_bwd(dout.contiguous(), idx, ctx.c)


class _ChannelGateFn(torch.autograd.Function):
    """fp32 [B, C] gate of ChannelGate (model.py:262-268): avg pool, max pool, the shared two-layer MLP, sigmoid."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        avg = ops.gap_fwd(x)
        mx, idx = ops.plane_max_fwd(x)
        w1f, w2f = w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1)
        cw, hid = ops.chan_gate_fwd(avg, mx, w1f, w2f)
        if _grad_mode() and any(ctx.needs_input_grad):
            ctx.save_for_backward(x, avg, mx, idx, w1, w2, cw, hid)
            ctx.mg = _main_grads((w1, w2))
        return cw

    @staticmethod
    def backward(ctx, dcw):
        x, avg, mx, idx, w1, w2, cw, hid = ctx.saved_tensors
        acc = ctx.mg is not None
        dw1, dw2 = ctx.mg if acc else (torch.empty_like(w1), torch.empty_like(w2))
        davg, dmx = ops.chan_gate_bwd(avg, mx, w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1), cw, hid,
                                      dcw.contiguous(), dw1, dw2, acc)
        dx = ops.pool_pair_bwd(davg, dmx, idx, x)
        return dx, (None if acc else dw1), (None if acc else dw2)


class _RefineMixFn(torch.autograd.Function):
    """low * sigmoid(s0 + s1) + high * cw (model.py:284-288)."""

    @staticmethod
    def forward(ctx, low, high, s, cw):
        if _grad_mode() and any(ctx.needs_input_grad):
            ctx.save_for_backward(low, high, s, cw)
        return ops.refine_mix_fwd(low, high, s, cw)

    @staticmethod
    def backward(ctx, dout):
        low, high, s, cw = ctx.saved_tensors
        return ops.refine_mix_bwd(low, high, s, cw, dout.contiguous())


class _ScaleAddFn(torch.autograd.Function):
    """a * para1 + y * para2 (model.py:331)."""

    @staticmethod
    def forward(ctx, a, y, p1, p2):
        if _grad_mode() and any(ctx.needs_input_grad):
            ctx.save_for_backward(a, y, p1, p2)
            ctx.mg = _main_grads((p1, p2))
        return ops.scale_add_fwd(a, y, p1.reshape(-1), p2.reshape(-1))

    @staticmethod
    def backward(ctx, dout):
        a, y, p1, p2 = ctx.saved_tensors
        acc = ctx.mg is not None
        dp1, dp2 = ctx.mg if acc else (torch.empty_like(p1), torch.empty_like(p2))
        da, dy = ops.scale_add_bwd(a, y, p1.reshape(-1), p2.reshape(-1), dout.contiguous(), dp1, dp2, acc)
        return da, dy, (None if acc else dp1), (None if acc else dp2)


# ----------------------------------------------------------------------------------------------- modules (reference interface)
class SpatialGate(nn.Module):
    """model.py:230-245."""

    def __init__(self):
        super().__init__()
        self.spatial = nn.Conv2d(2, 1, kernel_size=7, padding=3, bias=False)

    def planes(self, x):
        """The two per-channel halves of the 7x7 conv over [max_c x, mean_c x]; their sum is the conv's output (pre-sigmoid)."""
        return _apply(_DwConvFn, _apply(_MaxMeanFn, x), self.spatial.weight.view(2, 1, 7, 7), None)

    def forward(self, x):
        return torch.sigmoid(self.planes(x).sum(1, keepdim=True))      # (stand-alone use only: FreRefine fuses sum + sigmoid)


class ChannelGate(nn.Module):
    """model.py:248-268."""

    def __init__(self, dim):
        super().__init__()
        self.avg = nn.AdaptiveAvgPool2d((1, 1))
        self.max = nn.AdaptiveMaxPool2d((1, 1))
        self.mlp = nn.Sequential(nn.Conv2d(dim, dim // 16, 1, bias=False), nn.ReLU(), nn.Conv2d(dim // 16, dim, 1, bias=False))

    def gate(self, x):
        return _apply(_ChannelGateFn, x, self.mlp[0].weight, self.mlp[2].weight)             # fp32 [B, C]

    def forward(self, x):
        return self.gate(x).to(x.dtype).view(x.shape[0], x.shape[1], 1, 1)


class FreRefine(nn.Module):
    """model.py:273-290."""

    def __init__(self, dim):
        super().__init__()
        self.SpatialGate = SpatialGate()
        self.ChannelGate = ChannelGate(dim)
        self.proj = nn.Conv2d(dim, dim, kernel_size=1)

    def forward(self, low, high):
        mix = _apply(_RefineMixFn, low, high, self.SpatialGate.planes(high), self.ChannelGate.gate(low))
        return _apply(_Conv1x1Fn, mix, None, self.proj.weight, self.proj.bias)


class FreModule(nn.Module):
    """Adaptive frequency learning block (model.py:295-372).  ``conv`` and ``score_gen`` are registered, as in the reference,
    for checkpoint compatibility; its forward never uses them."""

    def __init__(self, dim, num_heads, bias, in_dim=3):
        super().__init__()
        self.conv = nn.Conv2d(in_dim, dim, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv1 = nn.Conv2d(in_dim, dim, kernel_size=3, stride=1, padding=1, bias=False)
        self.score_gen = nn.Conv2d(2, 2, 7, padding=3)
        self.para1 = nn.Parameter(torch.zeros(dim, 1, 1))
        self.para2 = nn.Parameter(torch.ones(dim, 1, 1))
        self.channel_cross_l = Chanel_Cross_Attention(dim, num_head=num_heads, bias=bias)
        self.channel_cross_h = Chanel_Cross_Attention(dim, num_head=num_heads, bias=bias)
        self.channel_cross_agg = Chanel_Cross_Attention(dim, num_head=num_heads, bias=bias)
        self.frequency_refine = FreRefine(dim)
        self.rate_conv = nn.Sequential(nn.Conv2d(dim, dim // 8, 1, bias=False), nn.GELU(), nn.Conv2d(dim // 8, 2, 1, bias=False))

    def fft(self, x, n=128):
        """(high, low) of the conv1 features of the resized image ``x`` (model.py:343-372)."""
        feat = _conv2d(x, self.conv1)
        H, W = feat.shape[-2:]
        half = None
        if H // n > 0 and W // n > 0:                    # below n pixels the reference's rectangle is empty by construction
            if max(H, W) > ops.L.lib().mi_fre_split_max_hw():
                raise RuntimeError(f"FreModule: feature maps above {ops.L.lib().mi_fre_split_max_hw()} pixels are not covered")
            with torch.no_grad():                        # .int() in the reference: no gradient reaches rate_conv
                w0, w2 = self.rate_conv[0].weight, self.rate_conv[2].weight
                half = ops.fre_rect(ops.gap_fwd(feat), w0.reshape(w0.shape[0], -1), w2.reshape(2, -1), H, W, n)
        return _apply(_FreSplitFn, feat, half)

    def forward(self, x, y):
        _, _, H, W = y.size()
        with torch.no_grad():
            x = ops.box_down(x.to(y.dtype).contiguous(), H, W)
        high_feature, low_feature = self.fft(x)
        high_feature = self.channel_cross_l(high_feature, y)
        low_feature = self.channel_cross_h(low_feature, y)
        agg = self.frequency_refine(low_feature, high_feature)
        out = self.channel_cross_agg(y, agg)
        return _apply(_ScaleAddFn, out, y, self.para1, self.para2)


class AdaIR(nn.Module):
    """The AdaIR network (model.py:378-496): the Restormer U-Net with a FreModule after the latent stage and after decoder
    levels 3 and 2; same constructor arguments, registration order and state_dict keys as the reference."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', decoder=True):
        super().__init__()
        c1, c2, c3, c4 = (int(dim * 2 ** i) for i in range(4))

        def stage(c, h, n):
            return _stage(c, h, n, ffn_expansion_factor, bias, LayerNorm_type)

        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.decoder = decoder
        if decoder:                                   # the three frequency modules all take heads[2] (model.py:402-404)
            for name, c in (("fre1", c4), ("fre2", c3), ("fre3", c2)):
                setattr(self, name, FreModule(c, num_heads=heads[2], bias=bias))
        self.encoder_level1, self.down1_2 = stage(c1, heads[0], num_blocks[0]), Downsample(c1)
        self.encoder_level2, self.down2_3 = stage(c2, heads[1], num_blocks[1]), Downsample(c2)
        self.encoder_level3, self.down3_4 = stage(c3, heads[2], num_blocks[2]), Downsample(c3)
        self.latent = stage(c4, heads[3], num_blocks[3])
        self.up4_3 = Upsample(c4)
        self.reduce_chan_level3 = nn.Conv2d(c4, c3, kernel_size=1, bias=bias)
        self.decoder_level3 = stage(c3, heads[2], num_blocks[2])
        self.up3_2 = Upsample(c3)
        self.reduce_chan_level2 = nn.Conv2d(c3, c2, kernel_size=1, bias=bias)
        self.decoder_level2 = stage(c2, heads[1], num_blocks[1])
        self.up2_1 = Upsample(c2)
        self.decoder_level1 = stage(c2, heads[0], num_blocks[0])
        self.refinement = stage(c2, heads[0], num_refinement_blocks)
        self.output = nn.Conv2d(c2, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)

    def forward(self, inp_img, noise_emb=None):
        def merge(up, deep, skip, reduce):            # concat-free: the two halves are the K panels of one 1x1 GEMM
            return _apply(_Conv1x1Fn, up(deep), skip, reduce.weight, reduce.bias)

        e1 = self.encoder_level1(self.patch_embed(inp_img))
        e2 = self.encoder_level2(self.down1_2(e1))
        e3 = self.encoder_level3(self.down2_3(e2))
        z = self.latent(self.down3_4(e3))
        if self.decoder:
            z = self.fre1(inp_img, z)
        z = self.decoder_level3(merge(self.up4_3, z, e3, self.reduce_chan_level3))
        if self.decoder:
            z = self.fre2(inp_img, z)
        z = self.decoder_level2(merge(self.up3_2, z, e2, self.reduce_chan_level2))
        if self.decoder:
            z = self.fre3(inp_img, z)
        z = self.refinement(self.decoder_level1(_up_cat(self.up2_1, z, e1)))
        return _conv2d(z, self.output, inp_img)
