"""Drop-in replacements for the block classes of the reference's ``AdaIR-main/net/model.py``.

``TransformerBlock`` / ``LayerNorm`` / ``FeedForward`` / ``Attention`` are the Restormer blocks (identical interface,
model.py:25-172).  ``Chanel_Cross_Attention`` (model.py:177-216) is MDTA with q from the image features ``x`` and k,v
from the frequency features ``y`` (3x3 depthwise on both branches)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .restormer import (Attention, FeedForward, LayerNorm, TransformerBlock, _apply, _CrossAttentionFn)  # noqa: F401

__all__ = ["Attention", "FeedForward", "LayerNorm", "TransformerBlock", "Chanel_Cross_Attention"]


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
