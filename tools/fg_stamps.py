"""Where does a wave of the fused GDFN kernel (fourth form, csrc/fused_gdfn.hip fg4_fwd_kernel) spend its cycles?  Runs the
STAMP build (MI_FG_DEBUG=0x1000: shader-clock stamps at the phase boundaries) and prints each phase's share of the wave's
lifetime.  python tools/fg_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_FG_DEBUG"] = str(0x1000)
import torch
import image_restoration_amd as m
from image_restoration_amd import ops

B = int(os.environ.get("BF_BATCH", "32"))
for C, H, W, h in ((48, 256, 256, 127), (96, 128, 128, 255), (96, 256, 256, 255)):
    torch.manual_seed(0)
    y = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16)
    ln_w, ln_b = 1 + 0.1 * torch.randn(C, device="cuda"), 0.1 * torch.randn(C, device="cuda")
    params = (torch.randn(2 * h, C, 1, 1, device="cuda") / C ** 0.5, None, torch.randn(2 * h, 1, 3, 3, device="cuda") / 3, None,
              torch.randn(C, h, 1, 1, device="cuda") / h ** 0.5, None)
    pack = ops.gdfn_fused_pack(y, ln_w, ln_b, params)
    if os.environ.get("FG_SAVE"):
        out, saved, mean, rstd = ops.gdfn_fused_fwd_train(y, pack, h, True)
    else:
        out, mean, rstd = ops.gdfn_fused_fwd(y, pack, h, True, want_stats=True)
    torch.cuda.synchronize()
    st = mean.flatten()[: 256 * 8 * 8].view(-1, 8)
    st = st[st[:, 6] > 0]
    tot = st[:, 6].mean().item()
    names = ["top barrier+stage", "LayerNorm", "barrier (+DMA wait)", "GEMM1 of group 0", "barriers", "phases: GEMM1(g+1) + conv/gate/GEMM2(g)"]
    parts = ", ".join(f"{n} {100 * st[:, i].mean().item() / tot:4.1f}%" for i, n in enumerate(names))
    acc = sum(st[:, i].mean().item() for i in range(6))
    print(f"C={C} {H}x{W} h={h} bs={B}: wave lifetime {tot:9.0f} cycles for {st[:, 7].mean().item():.0f} tiles = "
          f"{tot / st[:, 7].mean().item():7.0f} cycles/tile | {parts} | epilogue + rest {100 * (tot - acc) / tot:4.1f}%")
