"""Micro-benchmark of the one-launch backward tail (csrc/bwd_tail.hip) against the three kernels it replaces (weight-gradient
Gram, input-gradient GEMM, LayerNorm backward) at the Restormer-base training planes (BT_BATCH images, bf16).
Run on the GPU box: python tools/bench_tail.py   (BT_ABLATE=1 adds the MI_BT_DEBUG ablations)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_restoration_amd as m  # noqa: E402
from image_restoration_amd import ops  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("BT_BATCH", "32"))
SHAPES = [(48, 144, 256, 256), (48, 254, 256, 256), (96, 288, 128, 128), (96, 510, 128, 128), (96, 288, 256, 256),
          (96, 510, 256, 256)]


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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    only = os.environ.get("BT_ONLY")
    for C, M, H, W in SHAPES:
        if only and f"{C}x{M}x{H}" not in only.split(","):
            continue
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn((B, C, H, W), generator=g).to(DEV).to(torch.bfloat16)
        dy = torch.randn((B, M, H, W), generator=g).to(DEV).to(torch.bfloat16)
        dres = torch.randn((B, C, H, W), generator=g).to(DEV).to(torch.bfloat16)
        w = (0.1 * torch.randn((M, C), generator=g)).to(DEV)
        gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        xn, mean, rstd = ops.ln_fwd(x, gamma, beta, True, want_stats=True)
        dw, dg, db = torch.zeros((M, C), device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)

        def chain():
            ops.gram(dy, xn, sum_batch=True)                                   # weight gradient
            dxn = ops.conv1x1(dy, w, transposed=True)                          # input gradient
            return ops.ln_bwd(dxn, x, gamma, mean, rstd, dres, True, dg, db, False)

        def tail():
            return ops.bwd_tail(dy, x, dres, mean, rstd, w, gamma, beta, dw, dg, db, False)

        n = B * H * W
        alg = (M + 3 * C) * n * 2
        tt = timeit(tail)
        try:
            tc = timeit(chain)
        except Exception as e:  # the chain helper is only a yardstick
            tc = float("nan")
            print("chain failed:", e)
        print(f"tail C={C} M={M} {H}x{W} bs={B}: fused {tt:8.1f} us ({alg / tt / 1e6:6.2f} TB/s alg)   chain {tc:8.1f} us   "
              f"speed-up {tc / tt:.2f}x", flush=True)
        if os.environ.get("BT_ABLATE"):
            flags = ((1, "no wgrad"), (2, "no dxn"), (4, "no ds_add"), (8, "no LN phase"), (16, "no dx store"),
                     (32, "no compute"), (63, "staging only"), (64, "unrotated adds"))
            if os.environ.get("BT_ABLATE") != "1":
                flags = tuple((int(f), "dbg " + f) for f in os.environ["BT_ABLATE"].split(","))
            for flag, what in flags:
                os.environ["MI_BT_DEBUG"] = str(flag)
                m.reload_env()
                print(f"      {what:14s} {timeit(tail):8.1f} us", flush=True)
            os.environ["MI_BT_DEBUG"] = "0"
            m.reload_env()


if __name__ == "__main__":
    main()
