"""Where does a wave of the fused MDTA pass A kernel spend its cycles?  Runs the kernel with MI_FM_DEBUG=0x1000 (shader-clock
stamps around the phases; csrc/fused_mdta.hip) and prints each phase's share of the wave's lifetime.  Shares only: the stamps
serialise the waves' own overlap.  python tools/fm_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_FM_DEBUG"] = str(0x1000 | int(os.environ.get("FM_ABL", "0")))
import torch
import image_restoration_amd as m
from image_restoration_amd import ops

B = int(os.environ.get("BF_BATCH", "32"))
for C, heads, H, W in ((48, 1, 256, 256), (96, 2, 128, 128), (96, 1, 256, 256)):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16)
    ln_w, ln_b = 1 + 0.1 * torch.randn(C, device="cuda"), 0.1 * torch.randn(C, device="cuda")
    att = (torch.ones(heads, 1, 1, device="cuda"), torch.randn(3 * C, C, 1, 1, device="cuda") / C ** 0.5, None,
           torch.randn(3 * C, 1, 3, 3, device="cuda") / 3, None, torch.randn(C, C, 1, 1, device="cuda") / C ** 0.5, None)
    pack = ops.mdta_fused_pack(x, heads, ln_w, ln_b, att)
    out, mean, rstd = ops.mdta_fused_fwd(x, pack, att, heads, True, x, want_stats=True)
    torch.cuda.synchronize()
    nwg = min(256 // B if B <= 256 else 1, (H // 8) * (W // 32) // 4) * B if B <= 256 else B
    st = mean.flatten()[: 256 * 8 * 8].view(-1, 8)
    st = st[st[:, 6] > 0]
    tot = st[:, 6].mean().item()
    names = (["top barrier+stage", "LayerNorm", "barrier", "GEMM1", "barrier", "wave-local conv/Gram/stores"] if os.environ.get("MI_FM_CFG", "") == ""
             and C == 48 else ["stage+barriers", "LayerNorm", "GEMM1", "barrier wait", "Gram", "conv"])
    parts = ", ".join(f"{n} {100 * st[:, i].mean().item() / tot:4.1f}%" for i, n in enumerate(names))
    acc = sum(st[:, i].mean().item() for i in range(6))
    print(f"C={C} heads={heads} {H}x{W} bs={B}: wave lifetime {tot:9.0f} cycles for {st[:, 7].mean().item():.0f} tiles = "
          f"{tot / st[:, 7].mean().item():7.0f} cycles/tile | {parts} | unaccounted {100 * (tot - acc) / tot:4.1f}%")
