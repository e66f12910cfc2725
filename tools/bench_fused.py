"""Micro-benchmark of the fused half-block kernels against the unfused kernel chain they replace, at the Restormer-base
training planes (BF_BATCH images, bf16).  Run on the GPU box: python tools/bench_fused.py [gdfn|mdta]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_restoration_amd import ops  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("BF_BATCH", "32"))
SHAPES = [(48, 256, 256, 127), (96, 128, 128, 255), (96, 256, 256, 255)]


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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def gdfn():
    for C, H, W, h in SHAPES:
        torch.manual_seed(0)
        y = torch.randn(B, C, H, W, device=DEV).to(torch.bfloat16)
        ln_w = 1 + 0.1 * torch.randn(C, device=DEV)
        ln_b = 0.1 * torch.randn(C, device=DEV)
        params = (torch.randn(2 * h, C, 1, 1, device=DEV) / C ** 0.5, None, torch.randn(2 * h, 1, 3, 3, device=DEV) / 3, None,
                  torch.randn(C, h, 1, 1, device=DEV) / h ** 0.5, None)
        pack = ops.gdfn_fused_pack(y, ln_w, ln_b, params)
        fused = lambda: ops.gdfn_fused_fwd(y, pack, h, True, want_stats=True)

        def chain():
            yn, _, _ = ops.ln_fwd(y, ln_w, ln_b, True, want_stats=True)
            ops.gdfn_fwd(yn, y, params, True)

        if os.environ.get("BF_ABLATE"):
            for name, flag in (("full", 0), ("no GEMM1", 32), ("no conv", 64), ("no GEMM2", 128), ("no GEMM1/conv", 96),
                               ("only pro/epilogue", 224), ("prologue only", 1)):
                os.environ["MI_FG_DEBUG"] = str(flag)
                print(f"   ablation C={C} {H}x{W}: {name:20s} {timeit(fused):8.1f} us", flush=True)
            os.environ["MI_FG_DEBUG"] = "0"
        tf, tc = timeit(fused), timeit(chain)
        nbytes = 2.0 * B * C * H * W * 2
        flops = 2.0 * B * H * W * 3 * C * h
        print(f"gdfn fwd C={C} {H}x{W} h={h} bs={B}: fused {tf:8.1f} us ({nbytes / tf / 1e6:6.2f} TB/s alg, "
              f"{flops / tf / 1e6:6.1f} TF/s)   chain {tc:8.1f} us   speed-up {tc / tf:4.2f}x", flush=True)


def gdfn_train():
    """Training forward of the LN + GDFN half-block: one launch that also writes h0 and g (mi_gdfn_fused_fwd_train, incl. its
    per-step weight pack) against the chain mi_gdfn_fwd_ln (LN inside the project_in GEMM -> depthwise gate -> project_out)."""
    for C, H, W, h in SHAPES:
        torch.manual_seed(0)
        y = torch.randn(B, C, H, W, device=DEV).to(torch.bfloat16)
        ln_w = 1 + 0.1 * torch.randn(C, device=DEV)
        ln_b = 0.1 * torch.randn(C, device=DEV)
        params = (torch.randn(2 * h, C, 1, 1, device=DEV) / C ** 0.5, None, torch.randn(2 * h, 1, 3, 3, device=DEV) / 3, None,
                  torch.randn(C, h, 1, 1, device=DEV) / h ** 0.5, None)
        pack = ops.gdfn_fused_pack(y, ln_w, ln_b, params)
        fused = lambda: ops.gdfn_fused_fwd_train(y, ops.gdfn_fused_pack(y, ln_w, ln_b, params), h, True)
        infer = lambda: ops.gdfn_fused_fwd(y, pack, h, True, want_stats=True)
        chain = lambda: ops.gdfn_fwd(y, y, params, True, ln=(ln_w, ln_b, True))
        tf, ti, tc = timeit(fused), timeit(infer), timeit(chain)
        unit = 2.0 * B * C * H * W
        print(f"gdfn TRAINING fwd C={C} {H}x{W} h={h} bs={B}: one launch + saves {tf:8.1f} us ({(2 + 3.0 * h / C) * unit / tf / 1e6:5.2f} TB/s "
              f"alg)   same kernel without the saves {ti:8.1f} us   chain {tc:8.1f} us ({(5 + 6.0 * h / C) * unit / tc / 1e6:5.2f} TB/s alg)   "
              f"speed-up {tc / tf:4.2f}x", flush=True)


def mdta():
    """Fused pass A (LN -> qkv -> dw3x3 -> q k^T partials + v) + partial sum + fold + M v, against the unfused chain
    (qkv GEMM with LN on load, dw3x3, streaming Gram, fold, M v)."""
    import image_restoration_amd as m
    shapes = ((48, 1, 256, 256), (96, 2, 128, 128), (96, 1, 256, 256))
    if os.environ.get("BF_DEEP"):                         # the levels the fused pass does not cover: the chain's own numbers
        shapes = shapes + ((192, 4, 64, 64), (384, 8, 32, 32))
    for C, heads, H, W in shapes:
        torch.manual_seed(0)
        x = torch.randn(B, C, H, W, device=DEV).to(torch.bfloat16)
        ln_w = 1 + 0.1 * torch.randn(C, device=DEV)
        ln_b = 0.1 * torch.randn(C, device=DEV)
        att = (torch.ones(heads, 1, 1, device=DEV), torch.randn(3 * C, C, 1, 1, device=DEV) / C ** 0.5, None,
               torch.randn(3 * C, 1, 3, 3, device=DEV) / 3, None, torch.randn(C, C, 1, 1, device=DEV) / C ** 0.5, None)
        chain = lambda: ops.mdta_fwd(x, x, att, heads, False, ln=(ln_w, ln_b, False)) if ops.mdta_fwd_ln_ok(x, heads, 3) else \
            ops.mdta_fwd(ops.ln_fwd(x, ln_w, ln_b, True, want_stats=False)[0] if False else ops.ln_fwd(x, ln_w, ln_b, True)[0], x, att, heads, False)
        if not ops.mdta_fused_ok(x, heads, 3):
            N = float(B) * H * W
            tc = timeit(chain)
            print(f"mdta fwd C={C} heads={heads} {H}x{W} bs={B}: no fused kernel (tiles of 8 x 32 per CU: {B * (H // 8) * (W // 32) / 256:.1f}); "
                  f"chain {tc:8.1f} us = {2.0 * N * (4.0 * C * C + 2.0 * C * (C / heads) + 27.0 * C) / tc / 1e6:6.1f} TF/s over the half-block", flush=True)
            continue
        pack = ops.mdta_fused_pack(x, heads, ln_w, ln_b, att)
        fused = lambda: ops.mdta_fused_fwd(x, pack, att, heads, True, x)
        if os.environ.get("BF_ABLATE"):
            for name, flag in (("full", 0), ("no GEMM1", 32), ("no conv", 64), ("no Gram", 128), ("no GEMM1/conv", 96),
                               ("prologue + LN only", 224)):
                os.environ["MI_FM_DEBUG"] = str(flag)
                m.reload_env()
                print(f"   ablation C={C} heads={heads} {H}x{W}: {name:20s} {timeit(fused):8.1f} us (whole half-block)", flush=True)
            os.environ["MI_FM_DEBUG"] = "0"
            m.reload_env()
        ops.prof_enable(True)
        for _ in range(5):
            fused()
        torch.cuda.synchronize()
        tab = ops.prof_collect()
        ops.prof_enable(False)
        ka = tab["mdta_fused_a"]
        t_a = ka["ms"] / ka["launches"] * 1e3
        tf, tc = timeit(fused), timeit(chain)
        N = float(B) * H * W
        fl = 2.0 * N * (3.0 * C * C + C * (C / heads))
        print(f"mdta fwd C={C} heads={heads} {H}x{W} bs={B}: half-block fused {tf:8.1f} us  chain {tc:8.1f} us  speed-up {tc / tf:4.2f}x | "
              f"pass A kernel alone {t_a:8.1f} us = {2.0 * N * C * 2 / t_a / 1e6:5.2f} TB/s algorithmic (x in, v out), "
              f"{fl / t_a / 1e6:6.1f} TF/s MFMA ({fl / t_a / 1e6 / 2500 * 100:4.1f} % of the bf16 peak); "
              f"others: " + ", ".join(f"{k} {v['ms'] / 5 * 1e3:.0f} us" for k, v in tab.items() if k != "mdta_fused_a"), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "gdfn"
    {"gdfn": gdfn, "gdfn_train": gdfn_train, "mdta": mdta}[which]()
