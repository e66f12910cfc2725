"""Full-image tiled inference throughput (BASELINE configs[4]): Restormer base, 1024 x 1024, bf16, 224 + 2 x 16 tiles, then the same
with fp8 (e4m3) MFMA operands in the 1x1 projections and its PSNR bar against the bf16 and fp32 outputs.
python tools/bench_infer.py [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_restoration_amd as m  # noqa: E402
from image_restoration_amd import configs, inference, metrics, restormer  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
torch.manual_seed(0)
net = m.Restormer(**configs.RESTORMER_BASE).to(dev)
clean = torch.rand((1, 3, size, size), device=dev)
noisy = torch.clamp(torch.round(clean * 255) + 25 * torch.randn_like(clean), 0, 255) / 255
for fused in (2, 1, 0):
    os.environ["MI_NO_FUSED_INFER"] = "" if fused else "1"
    os.environ["MI_NO_FUSED_MDTA"] = "" if fused == 2 else "1"
    m.reload_env()
    for tb in (8, 25):
        out = inference.tiled_restore(net, noisy, tile_batch=tb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            out = inference.tiled_restore(net, noisy, tile_batch=tb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        p, s, _ = metrics.compute_psnr_ssim(out, clean)
        print(f"tiled inference {size}x{size}, 224+2x16 tiles, tile_batch {tb}, fused {['none', 'LN+GDFN', 'LN+GDFN and MDTA pass A'][fused]}: "
              f"{dt * 1e3:8.1f} ms/image = {size * size / dt / 1e6:6.2f} Mpix/s   (random-init net: PSNR {p:.2f} dB, SSIM {s:.4f}; "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)


def timed(tb):
    out = inference.tiled_restore(net, noisy, tile_batch=tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = inference.tiled_restore(net, noisy, tile_batch=tb)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / 3


def psnr(a, b):
    return float(10 * torch.log10(1.0 / torch.mean((a.double() - b.double()) ** 2)))


os.environ["MI_NO_FUSED_INFER"] = ""
os.environ["MI_NO_FUSED_MDTA"] = ""
m.reload_env()
with inference.PackedWeights(net):
    outp, dtp = timed(25)
out0, dt0 = timed(25)
print(f"packed 1x1 weight images kept across calls (inference.PackedWeights), tile_batch 25: {dtp * 1e3:.1f} ms/image = "
      f"{size * size / dtp / 1e6:.2f} Mpix/s  (per-call packing: {dt0 * 1e3:.1f} ms; outputs identical: {bool(torch.equal(outp, out0))})", flush=True)
ref32 = inference.tiled_restore(net, noisy, tile_batch=4, dtype=None)            # fp32 activations: the parity path
out16, dt16 = timed(25)
inference.calibrate_fp8(net, noisy, max_tiles=8)
print(f"fp8 projections (e4m3 MFMA operands, bf16 activations in HBM, static scales from 8 calibration tiles); bf16: "
      f"{dt16 * 1e3:.1f} ms/image, PSNR vs the fp32 path {psnr(out16, ref32):.2f} dB, vs clean {psnr(out16, clean):.3f} dB", flush=True)
for mode in ("attn", "all"):
    restormer.fp8_projections(net, mode)
    restormer.F8_COUNTS.update(f8=0, bf16=0)
    out8, dt8 = timed(25)
    cnt = dict(restormer.F8_COUNTS)
    print(f"  fp8 mode {mode:4s}: {dt8 * 1e3:8.1f} ms/image = {size * size / dt8 / 1e6:6.2f} Mpix/s   projections on fp8 operands: "
          f"{cnt['f8']} of {cnt['f8'] + cnt['bf16']}   PSNR vs bf16 output {psnr(out8, out16):.2f} dB, vs the fp32 path "
          f"{psnr(out8, ref32):.2f} dB, vs clean {psnr(out8, clean):.3f} dB (bf16: {psnr(out16, clean):.3f})", flush=True)
restormer.fp8_projections(net, None)
