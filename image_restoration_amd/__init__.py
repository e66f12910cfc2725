"""MI355X-native Restormer / MoCE-IR transformer block (hand-written gfx950 HIP kernels behind a C-ABI).

Drop-in modules with the reference's class names, constructor arguments, parameter names and
``forward`` signatures live in :mod:`image_restoration_amd.restormer`; the kernels are reached
through ``libmi_restore.so`` (``include/mi_restore.h``).  There is no CPU fallback: calling a module
on a CPU tensor, or without the built library, raises.
"""
from . import _lib  # noqa: F401
from .ops import reload_env  # noqa: F401
from .restormer import (Attention, Downsample, FeedForward, LayerNorm, OverlapPatchEmbed, Restormer,  # noqa: F401
                        TransformerBlock, Upsample)

__all__ = ["Attention", "Downsample", "FeedForward", "LayerNorm", "OverlapPatchEmbed", "Restormer",
           "TransformerBlock", "Upsample", "reload_env"]
