"""Dense 3x3 glue convolutions of Restormer base at the training planes (BF_BATCH images, bf16): implicit-GEMM kernels
(csrc/conv3x3.hip: forward, data gradient, weight gradient) against the im2col / col2im route they replace (forward + backward
through restormer._conv2d with MI_NO_CONV3_IMPLICIT=1).  python tools/bench_conv3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_restoration_amd as m  # noqa: E402
from image_restoration_amd import ops, restormer  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("BF_BATCH", "32"))
SHAPES = [("patch_embed", 3, 48, 256), ("down1_2", 48, 24, 256), ("down2_3", 96, 48, 128), ("down3_4", 192, 96, 64),
          ("up4_3", 384, 768, 32), ("up3_2", 192, 384, 64), ("up2_1", 96, 192, 128), ("output", 96, 3, 256)]


def timeit(fn, iters=10, warm=2):
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


tot_i = tot_o = 0.0
for name, cin, cout, hw in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, cin, hw, hw, device=DEV).to(torch.bfloat16)
    dy = torch.randn(B, cout, hw, hw, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    gf = 2.0 * 9 * cin * cout * B * hw * hw / 1e9
    mb = (cin + cout) * B * hw * hw * 2 / 1e6
    tf = timeit(lambda: ops.conv3x3(x, w))
    td = timeit(lambda: ops.conv3x3(dy, w, transpose=True)) if name != "patch_embed" else 0.0
    tw = timeit(lambda: ops.conv3x3_wgrad(dy, x))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(DEV)

    def step():
        xx = x.detach().requires_grad_(name != "patch_embed")
        restormer._conv2d(xx, conv).backward(dy)
        conv.weight.grad = None
    os.environ["MI_NO_CONV3_IMPLICIT"] = "1"
    m.reload_env()
    t_old = timeit(step)
    os.environ["MI_NO_CONV3_IMPLICIT"] = ""
    m.reload_env()
    t_new = timeit(step)
    tot_i += t_new
    tot_o += t_old
    print(f"{name:12s} {cin:4d}->{cout:4d} @{hw:3d}^2 bs {B}: fwd {tf:7.1f} us ({gf / tf * 1e3:6.1f} TF/s, {mb / tf:5.2f} TB/s)  dgrad {td:7.1f} us "
          f"({gf / td * 1e3 if td else 0:6.1f} TF/s)  wgrad {tw:7.1f} us ({gf / tw * 1e3:6.1f} TF/s) | fwd+bwd through the module: implicit "
          f"{t_new:7.1f} us, im2col route {t_old:7.1f} us  ({t_old / t_new:4.2f}x)", flush=True)
print(f"sum over the eight glue convs, fwd + bwd: implicit {tot_i / 1e3:.2f} ms, im2col route {tot_o / 1e3:.2f} ms")
