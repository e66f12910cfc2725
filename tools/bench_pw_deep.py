"""1x1 projections of the deep U-Net levels (Restormer base, BF_BATCH images, bf16): the LDS-tiled kernel (csrc/pw_lds.hip)
against the wave-owned / chunked kernels it replaces (MI_NO_PW_LDS=1).  python tools/bench_pw_deep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_restoration_amd as m  # noqa: E402
from image_restoration_amd import ops  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("BF_BATCH", "32"))
# (name, M, K, plane, transposed, residual, calls per training step)
SHAPES = [("L4 qkv", 1152, 384, 32, False, False, 8), ("L4 attn out", 384, 384, 32, False, True, 8), ("L4 project_in", 2042, 384, 32, False, False, 8),
          ("L4 project_out", 384, 1021, 32, False, True, 8), ("L4 d project_out", 1021, 384, 32, True, False, 8),
          ("L4 d project_in", 384, 2042, 32, True, False, 8), ("L4 d qkv", 384, 1152, 32, True, False, 8),
          ("L3 qkv", 576, 192, 64, False, False, 12), ("L3 attn out", 192, 192, 64, False, True, 12), ("L3 project_in", 1020, 192, 64, False, False, 12),
          ("L3 project_out", 192, 510, 64, False, True, 12), ("L3 d project_out", 510, 192, 64, True, False, 12),
          ("L3 d project_in", 192, 1020, 64, True, False, 12), ("L3 d qkv", 192, 576, 64, True, False, 12),
          ("L2 project_out", 96, 255, 128, False, True, 12), ("L2 d project_in", 96, 510, 128, True, False, 12)]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


gain = 0.0
for name, M, K, hw, tr, res, calls in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, K, hw, hw, device=DEV).to(torch.bfloat16)
    w = torch.randn((K, M) if tr else (M, K), device=DEV) / K ** 0.5
    r = torch.randn(B, M, hw, hw, device=DEV).to(torch.bfloat16) if res else None
    y = torch.empty(B, M, hw, hw, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.conv1x1(x, w, None, r, transposed=tr, out=y)
    t = {}
    os.environ["MI_PW_LDS"] = "all"
    for mode in ("old", "lds"):
        os.environ["MI_NO_PW_LDS"] = "1" if mode == "old" else ""
        m.reload_env()
        t[mode] = timeit(fn)
    gf = 2.0 * M * K * B * hw * hw / 1e9
    mb = (M + K + (M if res else 0)) * B * hw * hw * 2 / 1e6
    gain += (t["old"] - t["lds"]) * calls if t["lds"] < t["old"] else 0.0
    print(f"{name:18s} M={M:5d} K={K:5d} @{hw:3d}^2 bs {B} {'W^T' if tr else '   '}: LDS-tiled {t['lds']:7.1f} us ({gf / t['lds'] * 1e3:6.1f} TF/s, "
          f"{mb / t['lds']:5.2f} TB/s)   before {t['old']:7.1f} us ({gf / t['old'] * 1e3:6.1f} TF/s)   {t['old'] / t['lds']:4.2f}x", flush=True)
print(f"sum of the gains where the LDS-tiled kernel wins, per training step: {gain / 1e3:.2f} ms")
