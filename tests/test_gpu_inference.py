"""GPU tests of the rows either side of the block path (BASELINE configs[4]): device-side sample
pipeline vs a numpy restatement of the reference's recipe, PSNR / SSIM vs the oracle, tiled inference tile-vs-tile."""
import numpy as np
import pytest
import torch

from oracle import restormer_ref as R
from oracle.fixtures import seeded_input

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _augment_numpy(a, mode):
    """MoCE-IR-main/src/utils/image_utils.py data_augmentation, mode numbering as the reference (HWC array)."""
    if mode == 0:
        return a
    if mode == 1:
        return np.flipud(a)
    out = np.rot90(a, k={2: 1, 3: 1, 4: 2, 5: 2, 6: 3, 7: 3}[mode])
    return np.flipud(out) if mode in (3, 5, 7) else out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_batch_matches_reference_recipe(dtype):
    """crop -> dihedral augmentation -> sigma noise on the uint8 grid -> ToTensor, all eight modes, ragged image sizes."""
    from image_restoration_amd import data as D
    rng = np.random.default_rng(5)
    pool = D.ImagePool(DEV)
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in ((70, 93), (64, 64), (101, 80))]
    for im in imgs:
        pool.add(im)
    pool.finalize()
    kept = [D.crop_to_multiple(im, 16) for im in imgs]
    assert [k.shape[:2] for k in kept] == [(64, 80), (64, 64), (96, 80)]
    B, P = 16, 48
    sample = torch.tensor([i % 3 for i in range(B)], dtype=torch.int32)
    top = torch.tensor([(5 * i) % (kept[i % 3].shape[0] - P + 1) for i in range(B)], dtype=torch.int32)
    left = torch.tensor([(7 * i) % (kept[i % 3].shape[1] - P + 1) for i in range(B)], dtype=torch.int32)
    mode = torch.tensor([i % 8 for i in range(B)], dtype=torch.int32)
    sig = torch.tensor([(15.0, 25.0, 50.0)[i % 3] for i in range(B)])
    noise = seeded_input((B, 3, P, P), 77)
    degraded, clean = D.patch_batch(pool, (sample, top, left, mode, sig), P, dtype, noise=noise.to(DEV))
    for b in range(B):
        patch = kept[int(sample[b])][int(top[b]):int(top[b]) + P, int(left[b]):int(left[b]) + P]
        aug = np.ascontiguousarray(_augment_numpy(patch, int(mode[b])))                     # HWC uint8
        ref_clean = torch.from_numpy(aug).permute(2, 0, 1).float() / 255.0                # ToTensor
        n_hwc = noise[b].permute(1, 2, 0).numpy().astype(np.float64)
        ref_deg = np.clip(aug.astype(np.float64) + n_hwc * float(sig[b]), 0, 255).astype(np.uint8)   # degradation_utils.py:21-24
        ref_deg = torch.from_numpy(ref_deg).permute(2, 0, 1).float() / 255.0
        assert torch.equal(clean[b].float().cpu(), ref_clean.to(dtype).float()), (b, int(mode[b]))
        assert torch.equal(degraded[b].float().cpu(), ref_deg.to(dtype).float()), (b, int(mode[b]))     # bit-exact uint8 grid


def test_draw_batch_plan_stays_inside_images():
    from image_restoration_amd import data as D
    pool = D.ImagePool(DEV)
    for h, w in ((64, 64), (80, 112)):
        pool.add(np.zeros((h, w, 3), dtype=np.uint8))
    pool.finalize()
    g = torch.Generator().manual_seed(0)
    sample, top, left, mode, sig = D.draw_batch_plan(pool, 256, 64, (15.0, 25.0, 50.0), g)
    hw = pool._hw[sample.long()]
    assert bool(((top >= 0) & (top.long() + 64 <= hw[:, 0]) & (left >= 0) & (left.long() + 64 <= hw[:, 1])).all())
    assert set(mode.tolist()) == set(range(1, 8)) and set(sig.tolist()) == {15.0, 25.0, 50.0}


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (3, 3, 40, 72)), (torch.bfloat16, (2, 3, 96, 64)), (torch.float32, (1, 1, 7, 9))])
def test_psnr_ssim_vs_oracle(dtype, shape):
    from image_restoration_amd import metrics
    clean = torch.rand(shape, generator=torch.Generator().manual_seed(1))
    rest = (clean + 0.1 * seeded_input(shape, 3)).to(dtype)                       # leaves [0,1]: the clipping matters
    clean = clean.to(dtype)
    psnr, ssim = metrics.psnr_ssim_per_image(rest.to(DEV), clean.to(DEV))
    for b in range(shape[0]):
        assert abs(float(psnr[b]) - R.psnr(rest[b:b + 1].float(), clean[b:b + 1].float())) < 1e-3
        assert abs(float(ssim[b]) - R.ssim(rest[b:b + 1].float(), clean[b:b + 1].float())) < 1e-4
    p, s, n = metrics.compute_psnr_ssim(rest.to(DEV), clean.to(DEV))
    assert n == shape[0] and abs(p - float(psnr.mean())) < 1e-5 and abs(s - float(ssim.mean())) < 1e-6


def test_tiled_inference_is_tile_exact_and_covers_ragged_images():
    """configs[4] parity is tile-vs-tile: every output cell equals the network run on that cell's padded window alone; the
    stitched image has the input's size also when it is not a multiple of the tile."""
    import image_restoration_amd as m
    import torch.nn.functional as F
    from image_restoration_amd import inference
    from image_restoration_amd.configs import RESTORMER_TINY
    net = m.Restormer(**RESTORMER_TINY)
    net.load_state_dict(R.make_restormer_state(RESTORMER_TINY, seed=1))
    net = net.to(DEV)
    img = torch.rand((1, 3, 200, 136), generator=torch.Generator().manual_seed(2)).to(DEV)
    tile, ov = 96, 16
    out = inference.tiled_restore(net, img, tile=tile, overlap=ov, tile_batch=3)
    assert out.shape == img.shape and out.dtype == img.dtype
    xp = F.pad(img.to(torch.bfloat16), (ov, ov + 192 - 136, ov, ov + 288 - 200), mode="replicate")
    with torch.no_grad():
        for (i, j) in ((0, 0), (1, 1), (2, 0)):
            win = xp[:, :, i * tile:i * tile + tile + 2 * ov, j * tile:j * tile + tile + 2 * ov].contiguous()
            ref = net(win)[:, :, ov:ov + tile, ov:ov + tile]
            h = min(tile, 200 - i * tile)
            w = min(tile, 136 - j * tile)
            got = out[:, :, i * tile:i * tile + h, j * tile:j * tile + w]
            assert float((got - ref[:, :, :h, :w].float()).abs().max()) < 2e-2     # batch composition only changes bf16 rounding paths


def test_packed_weights_cache_for_inference_follows_the_weights():
    """inference.PackedWeights (the library's packed 1x1 weight images kept across no_grad calls): bit-identical outputs with and
    without it; a load_state_dict and an in-place write are both picked up before the next forward (the stale-image hazard);
    closing it returns the library to per-call packing."""
    import image_restoration_amd as m
    from image_restoration_amd import inference, ops
    from image_restoration_amd.configs import RESTORMER_TINY
    net = m.Restormer(**RESTORMER_TINY)
    net.load_state_dict(R.make_restormer_state(RESTORMER_TINY, seed=3))
    net = net.to(DEV).eval()
    # (128 x 128: every level's planes are >= 16 rows, so the whole network runs on the native kernels - the 8 x 8 planes of a 64 x 64
    #  input would send the latent-level 3x3 convs through torch / MIOpen, whose results are not bit-reproducible from run to run)
    x = torch.rand((2, 3, 128, 128), generator=torch.Generator().manual_seed(4)).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        ref1 = net(x).float()
        assert torch.equal(net(x).float(), ref1)
        with inference.PackedWeights(net) as pk:
            a = net(x).float()                       # registers the matrices, packs per call
            assert ops.pw_cache_pending()
            b = net(x).float()                       # refreshed by the pre-hook, then served from the cache
            assert not ops.pw_cache_pending()
            assert torch.equal(a, ref1) and torch.equal(b, ref1)
            sd2 = R.make_restormer_state(RESTORMER_TINY, seed=5)
            net.load_state_dict(sd2)
            c = net(x).float()
            for p in net.parameters():               # an in-place write outside load_state_dict
                p.mul_(1.5)
                break
            d = net(x).float()
        e = net(x).float()                           # cache closed: per-call packing again
        ref2 = m.Restormer(**RESTORMER_TINY)
        ref2.load_state_dict(sd2)
        ref2 = ref2.to(DEV).eval()
        r2 = ref2(x).float()
        assert torch.equal(c, r2), float((c - r2).abs().max())
        assert not torch.equal(c, ref1)
        next(ref2.parameters()).mul_(1.5)
        want = ref2(x).float()
        assert torch.equal(d, want), float((d - want).abs().max())
        assert torch.equal(e, want), float((e - want).abs().max())
