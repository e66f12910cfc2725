"""Full-image tiled inference throughput (BASELINE configs[4]): Restormer base, 1024 x 1024, bf16, 224 + 2 x 16 tiles.
python tools/bench_infer.py [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_restoration_amd as m  # noqa: E402
from image_restoration_amd import configs, inference, metrics  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
torch.manual_seed(0)
net = m.Restormer(**configs.RESTORMER_BASE).to(dev)
clean = torch.rand((1, 3, size, size), device=dev)
noisy = torch.clamp(torch.round(clean * 255) + 25 * torch.randn_like(clean), 0, 255) / 255
for fused in (1, 0):
    os.environ["MI_NO_FUSED_INFER"] = "" if fused else "1"
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
        print(f"tiled inference {size}x{size}, 224+2x16 tiles, tile_batch {tb}, fused LN+GDFN {'on' if fused else 'off'}: "
              f"{dt * 1e3:8.1f} ms/image = {size * size / dt / 1e6:6.2f} Mpix/s   (random-init net: PSNR {p:.2f} dB, SSIM {s:.4f}; "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)
