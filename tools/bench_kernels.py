#!/usr/bin/env python3
"""Per-shape kernel microbenchmarks (run on the GPU box): python tools/bench_kernels.py [pw|ln|dw|gram]
Times each call with HIP events on the current stream, prints algorithmic GB/s."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bench_pw():
    from image_restoration_amd import ops
    B = int(os.environ.get("BK_BATCH", "8"))
    shapes = [  # (M, K, H, W, transposed, residual)
        (144, 48, 256, 256, False, False), (48, 48, 256, 256, False, True), (254, 48, 256, 256, False, False),
        (48, 127, 256, 256, False, True), (48, 144, 256, 256, True, False), (127, 48, 256, 256, True, False),
        (48, 254, 256, 256, True, False),
        (288, 96, 256, 256, False, False), (510, 96, 256, 256, False, False), (96, 255, 256, 256, False, True),
        (96, 288, 256, 256, True, False), (96, 510, 256, 256, True, False),
        (288, 96, 128, 128, False, False), (510, 96, 128, 128, False, False), (96, 255, 128, 128, False, True),
        (576, 192, 64, 64, False, False), (1020, 192, 64, 64, False, False), (1152, 384, 32, 32, False, False),
    ]
    if os.environ.get("BK_DEEP"):      # the C >= 192 levels only (forward, input-gradient and attention-product shapes)
        shapes = [(576, 192, 64, 64, False, False), (1020, 192, 64, 64, False, False), (192, 510, 64, 64, False, True),
                  (192, 192, 64, 64, False, True), (192, 576, 64, 64, True, False), (192, 1020, 64, 64, True, False),
                  (510, 192, 64, 64, True, False),
                  (1152, 384, 32, 32, False, False), (2042, 384, 32, 32, False, False), (384, 1021, 32, 32, False, True),
                  (384, 384, 32, 32, False, True), (384, 1152, 32, 32, True, False), (384, 2042, 32, 32, True, False),
                  (1021, 384, 32, 32, True, False)]
    for (M, K, H, W, tr, res) in shapes:
        x = torch.randn(B, K, H, W, device="cuda").bfloat16()
        w = torch.randn((K, M) if tr else (M, K), device="cuda")
        r = torch.randn(B, M, H, W, device="cuda").bfloat16() if res else None
        us = timeit(lambda: ops.conv1x1(x, w, None, r, tr))
        gb = (K + M + (M if res else 0)) * B * H * W * 2 / 1e9
        print(f"pw M={M:4d} K={K:4d} {H}x{W} tr={int(tr)} res={int(res)}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  "
              f"{2 * M * K * B * H * W / us / 1e6:7.1f} TF/s", flush=True)


def bench_ln():
    from image_restoration_amd import ops
    for (C, H, W) in [(48, 256, 256), (96, 256, 256), (96, 128, 128), (192, 64, 64), (384, 32, 32)]:
        B = int(os.environ.get("BK_BATCH", "8"))
        x = torch.randn(B, C, H, W, device="cuda").bfloat16()
        w, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        us = timeit(lambda: ops.ln_fwd(x, w, b, True))
        y, mean, rstd = ops.ln_fwd(x, w, b, True)
        print(f"ln_fwd C={C} {H}x{W}: {us:8.1f} us {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
        dw, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dy, dres = torch.randn_like(x), torch.randn_like(x)   # distinct tensors: three reads + one write of real traffic
        us = timeit(lambda: ops.ln_bwd(dy, x, w, mean, rstd, dres, True, dw, db, False))
        print(f"ln_bwd C={C} {H}x{W}: {us:8.1f} us {4 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)


def bench_dw():
    from image_restoration_amd import ops
    for (C, H, W) in [(144, 256, 256), (254, 256, 256), (288, 256, 256), (510, 128, 128), (1020, 64, 64)]:
        B = int(os.environ.get("BK_BATCH", "8"))
        x = torch.randn(B, C, H, W, device="cuda").bfloat16()
        w = torch.randn(C, 1, 3, 3, device="cuda")
        us = timeit(lambda: ops.dwconv_fwd(x, w, None))
        print(f"dw_fwd  C={C} {H}x{W}: {us:8.1f} us {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
        us = timeit(lambda: ops.dwconv_bwd(x, x, w, False))
        print(f"dw_bwd(data+wgrad) C={C} {H}x{W}: {us:8.1f} us {4 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
        if C % 2 == 0:
            us = timeit(lambda: ops.dwconv_gate_fwd(x, w, None))
            print(f"dw_gate_fwd C={C} {H}x{W}: {us:8.1f} us {2.5 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
            dg = x[:, : C // 2].contiguous()
            us = timeit(lambda: ops.dwconv_gate_bwd(dg, x, x, w, False))
            print(f"dw_gate_bwd C={C} {H}x{W}: {us:8.1f} us {3.5 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
            us = timeit(lambda: ops.dwconv_gate_bwd_recompute(dg, x, w, None))
            print(f"dw_gate_bwd_recompute C={C} {H}x{W}: {us:8.1f} us {2.5 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
            us = timeit(lambda: ops.dwconv_gate_fwd(x, w, None, want_y=False))
            print(f"dw_gate_fwd(no y) C={C} {H}x{W}: {us:8.1f} us {1.5 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)


def bench_gram():
    from image_restoration_amd import ops
    for (ma, mb, H, W, g, sb) in [(48, 48, 256, 256, 1, False), (96, 96, 256, 256, 1, False), (144, 48, 256, 256, 1, True),
                                  (254, 48, 256, 256, 1, True), (510, 96, 256, 256, 1, True), (48, 127, 256, 256, 1, True),
                                  (48, 48, 128, 128, 2, False), (1020, 192, 64, 64, 1, True),
                                  (288, 96, 256, 256, 1, True), (96, 255, 256, 256, 1, True), (288, 96, 128, 128, 1, True),
                                  (96, 255, 128, 128, 1, True), (96, 96, 128, 128, 1, True),
                                  # the C >= 192 levels' weight gradients (every one of them sums over the batch)
                                  (576, 192, 64, 64, 1, True), (192, 510, 64, 64, 1, True), (192, 192, 64, 64, 1, True),
                                  (1152, 384, 32, 32, 1, True), (2042, 384, 32, 32, 1, True), (384, 1021, 32, 32, 1, True),
                                  (384, 384, 32, 32, 1, True), (1728, 384, 32, 32, 1, True), (864, 192, 64, 64, 1, True)]:
        B = int(os.environ.get("BK_BATCH", "8"))   # 32: operands exceed the 256 MB Infinity Cache, like in the real step
        if os.environ.get("BK_DEEP") and H > 64:
            continue
        a = torch.randn(B, ma * g, H, W, device="cuda").bfloat16()
        b = torch.randn(B, mb * g, H, W, device="cuda").bfloat16()
        us = timeit(lambda: ops.gram(a, b, g, sb))
        print(f"gram {ma}x{mb} g={g} {H}x{W} sum_batch={int(sb)}: {us:8.1f} us {(a.numel() + b.numel()) * 2 / us / 1e3:7.0f} GB/s",
              flush=True)
        if not sb:
            us = timeit(lambda: ops.gram(a, b, g, False, True))
            print(f"gram+sumsq {ma}x{mb} g={g} {H}x{W}: {us:8.1f} us {(a.numel() + b.numel()) * 2 / us / 1e3:7.0f} GB/s", flush=True)


def bench_dqk():
    """The merged q / k gradient GEMM of MDTA: per-image [2c][2c] weights over the stacked [k; q], two outputs (y_split)."""
    import ctypes as C
    from image_restoration_amd import _lib as L, ops
    B = int(os.environ.get("BK_BATCH", "8"))
    for (c, heads, H, W) in [(48, 1, 256, 256), (96, 1, 256, 256), (48, 2, 128, 128), (48, 4, 64, 64), (48, 8, 32, 32)]:
        Cc, N = c * heads, H * W
        qkv = torch.randn(B, 3 * Cc, H, W, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv)
        w = torch.randn(B, heads, 2 * c, 2 * c, device="cuda")
        def desc(rows, row0, y_off, split):
            d = L.PwDesc()
            d.x1, d.x1_bs, d.x1_gs, d.k1 = qkv.data_ptr() + Cc * N * 2, 3 * Cc * N, c * N, c      # k
            d.x2, d.x2_bs, d.x2_gs, d.k2 = qkv.data_ptr(), 3 * Cc * N, c * N, c                    # q
            d.w = w.data_ptr() + row0 * 2 * c * 4
            d.w_bs, d.w_gs, d.w_sm, d.w_sk = heads * 4 * c * c, 4 * c * c, 2 * c, 1
            d.y, d.y_bs, d.y_gs = dqkv.data_ptr() + y_off * N * 2, 3 * Cc * N, c * N
            d.m, d.n, d.batch, d.groups, d.dtype = rows, N, B, heads, 1
            if split:
                d.y_split, d.y2, d.y2_bs, d.y2_gs = c, dqkv.data_ptr() + Cc * N * 2, 3 * Cc * N, c * N
            return d
        merged, dq, dk = desc(2 * c, 0, 0, True), desc(c, 0, 0, False), desc(c, c, Cc, False)
        t1 = timeit(lambda: ops.pw_gemm_desc(merged, qkv.device))
        t2 = timeit(lambda: (ops.pw_gemm_desc(dq, qkv.device), ops.pw_gemm_desc(dk, qkv.device)))
        print(f"dq/dk c={c} heads={heads} {H}x{W}: one GEMM, two outputs {t1:8.1f} us   two GEMMs {t2:8.1f} us", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pw", "ln", "dw", "gram"]
    for wname in which:
        {"pw": bench_pw, "ln": bench_ln, "dw": bench_dw, "gram": bench_gram, "dqk": bench_dqk}[wname]()

