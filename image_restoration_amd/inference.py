"""Full-image tiled inference (BASELINE configs[4]).

The reference's test loops push the whole image through the network (MoCE-IR-main/src/test.py:82-123, after cropping it to a
multiple of the base size, src/utils/image_utils.py:62-67); its tiling helpers (image_utils.py:71-101) are never called by a
harness, so the tiling below is this build's own definition:

  * the image is cut into ``tile`` x ``tile`` output cells; each cell is restored from the cell plus ``overlap`` pixels of
    context on every side (replicated at the image border, like the reference helper's ``np.pad(mode='edge')``), i.e. the
    network sees (tile + 2 overlap)^2 inputs and only the central tile^2 outputs are kept;
  * the cells of an image are independent network evaluations: they are batched ``tile_batch`` at a time through the
    no_grad path (fused LN + GDFN kernel, nothing saved for backward);
  * MDTA statistics (L2 norms over all pixels, C x C attention) are global over whatever the network sees, so tiled output
    is NOT the full-image output; parity for this configuration is tile-versus-tile (tests/test_gpu_inference.py).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def crop_to_multiple(img: torch.Tensor, base: int = 16) -> torch.Tensor:
    """image_utils.crop_img on a [B,C,H,W] tensor."""
    h, w = img.shape[-2:]
    ch, cw = h % base, w % base
    return img[..., ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2]


class PackedWeights:
    """Inference-time owner of the library's packed-weight cache (mi_pw_cache_*; the trainer owns it while training).

    Every 1x1 GEMM packs its fp32 weight matrix into a bf16 LDS image - per call, unless a caller that controls when the weights
    change lends the library a buffer.  Under ``no_grad`` they never change between calls: this object moves the model's
    parameters into one flat buffer (the cache only trusts matrices inside a range it was given - a temporary may reuse an
    address), lends the cache, and re-packs everything in ONE launch whenever new matrices were seen (the first forward) or the
    parameters were written (``load_state_dict``, any in-place update: the parameters' version counters).  Use as a
    context manager or call ``close()``: the cache is process-global."""

    def __init__(self, model):
        from . import ops
        params = list(model.parameters())
        dev = params[0].device
        if dev.type != "cuda" or any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise RuntimeError("PackedWeights: fp32 parameters on one MI355X")
        self._ops = ops
        self._params = params
        self._hooks = []
        self._token = None
        self._saved_data = None
        # A model a FlatTrainer owns (its parameters are views of the trainer's flat buffer and carry main_grad) already has a
        # cache owner that follows every weight update: per-epoch validation inside a training run borrows it.  Moving the
        # parameters into a second flat buffer here would cut them loose from the buffer AdamW updates - training would go on
        # "updating" memory the model no longer reads.
        self._borrowed = any(getattr(p, "main_grad", None) is not None for p in params)
        if self._borrowed:
            if ops.pw_cache_owner() is None:
                raise RuntimeError("PackedWeights: the model's parameters belong to a FlatTrainer whose packed-weight cache is "
                                   "closed; validate without PackedWeights or keep the trainer's cache (pack_cache=True)")
            self.flat = None
            self._open = True
            return
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._saved_data = [p.data for p in params]
        with torch.no_grad():
            for p, o in zip(params, offs):
                self.flat[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
        # (the cache is process-global: a previous owner - some other model's trainer - falls back to per-call weight packing,
        #  which is always correct; its own close() no longer matches the owner token and leaves this cache alone)
        self._token = ops.pw_cache_enable(int(total) * 8 + (4 << 20), dev, self.flat)
        self._version = self._weights_version()
        self._hooks = [model.register_load_state_dict_post_hook(lambda *_: self.refresh()),
                       model.register_forward_pre_hook(lambda *_: self._check())]
        self._open = True

    def _weights_version(self) -> int:
        # (views made through .data do not share the flat buffer's version counter: sum the parameters' own)
        return self.flat._version + sum(p._version for p in self._params)

    def refresh(self) -> None:
        if self._open and not self._borrowed:
            self._ops.bump_weights_epoch()
            self._ops.pw_cache_refresh()
            self._version = self._weights_version()

    def _check(self) -> None:
        if self._open and not self._borrowed and (self._weights_version() != self._version or self._ops.pw_cache_pending()):
            self.refresh()

    def close(self) -> None:
        if not self._open:
            return
        self._open = False
        if self._borrowed:
            return
        self._ops.pw_cache_release(self._token)
        self._token = None
        for h in self._hooks:
            h.remove()
        # hand the parameters back their own storage (current values): nothing keeps aliasing the flat buffer
        with torch.no_grad():
            for p, old in zip(self._params, self._saved_data):
                old.copy_(p.data)
                p.data = old
        self._saved_data = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cells(img: torch.Tensor, tile: int, overlap: int, dtype):
    B, C, H0, W0 = img.shape
    x = img.to(dtype) if dtype is not None else img
    H, W = -(-H0 // tile) * tile, -(-W0 // tile) * tile
    xp = F.pad(x, (overlap, overlap + W - W0, overlap, overlap + H - H0), mode="replicate")
    cells = [(b, i, j) for b in range(B) for i in range(H // tile) for j in range(W // tile)]
    return xp, cells, H, W


@torch.no_grad()
def calibrate_fp8(model, img: torch.Tensor, tile: int = 224, overlap: int = 16, max_tiles: int = 8,
                  dtype: Optional[torch.dtype] = torch.bfloat16) -> None:
    """Static activation scales for the fp8 projections (restormer.fp8_calibrate) from up to ``max_tiles`` cells spread over
    ``img``; afterwards ``restormer.fp8_projections(model, "all" | "attn")`` switches the no_grad forward over."""
    from . import restormer
    xp, cells, _, _ = _cells(img, tile, overlap, dtype)
    size = tile + 2 * overlap
    pick = cells[::max(1, len(cells) // max_tiles)][:max_tiles]
    batch = torch.stack([xp[b, :, i * tile:i * tile + size, j * tile:j * tile + size] for b, i, j in pick]).contiguous()
    was_training = model.training
    model.eval()
    try:
        restormer.fp8_calibrate(model, [batch])
    finally:
        model.train(was_training)


@torch.no_grad()
def tiled_restore(model, img: torch.Tensor, tile: int = 224, overlap: int = 16, tile_batch: int = 8,
                  dtype: Optional[torch.dtype] = torch.bfloat16) -> torch.Tensor:
    """img [B,3,H,W] in [0,1] -> restored [B,3,H,W] (same dtype as ``img``).  The default 224 + 2 x 16 gives 256 x 256
    network inputs: every level then has power-of-two rows (256 / 128 / 64 / 32), which is what the streaming depthwise, the
    native 3x3 glue and the fused LN + GDFN kernels are built for.  H, W need not be multiples of ``tile``: the image is
    edge-replicated up to the cell grid and the result cropped back."""
    B, C, H0, W0 = img.shape
    xp, cells, H, W = _cells(img, tile, overlap, dtype)
    size = tile + 2 * overlap
    out = torch.empty((B, C, H, W), dtype=xp.dtype, device=xp.device)
    was_training = model.training
    model.eval()
    try:
        for s in range(0, len(cells), tile_batch):
            chunk = cells[s:s + tile_batch]
            batch = torch.stack([xp[b, :, i * tile:i * tile + size, j * tile:j * tile + size] for b, i, j in chunk]).contiguous()
            y = model(batch)
            for k, (b, i, j) in enumerate(chunk):
                out[b, :, i * tile:(i + 1) * tile, j * tile:(j + 1) * tile] = y[k, :, overlap:overlap + tile, overlap:overlap + tile]
    finally:
        model.train(was_training)
    return out[:, :, :H0, :W0].to(img.dtype)
