"""Where does a wave of the backward-tail kernel spend its cycles?  Needs the -DBT_STAMP variant of the library
(make BUILD=build_st LIB=../libmi_restore_btstamp.so EXTRA=-DBT_STAMP; MI_RESTORE_LIB=.../libmi_restore_btstamp.so): shader-clock
time per phase, summed per wave, written in place of the [G | S] partials.  Shares only: the stamps cost a few cycles each.
python tools/bt_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as m
from image_restoration_amd import ops, _lib as L

NAMES = ["sync4 wait+loop", "vmcnt(0) wait", "stage dy0 + normalise x", "sync1 wait", "issue dy1 + LN(prev)", "sync2 wait",
         "store(prev)", "wgrad half 0", "dy1 wait + stage", "issue dy0(next) + wgrad half 1", "sync3 wait", "issue x(next) + dxn"]
B = int(os.environ.get("BT_BATCH", "32"))
for C, M, H, W, NW in ((96, 510, 256, 256, 8), (96, 288, 256, 256, 8), (48, 254, 256, 256, 4)):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((B, C, H, W), generator=g).cuda().bfloat16()
    dy = torch.randn((B, M, H, W), generator=g).cuda().bfloat16()
    dres = torch.randn((B, C, H, W), generator=g).cuda().bfloat16()
    w = (0.1 * torch.randn((M, C), generator=g)).cuda()
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    _, mean, rstd = ops.ln_fwd(x, gamma, beta, True, want_stats=True)
    dw, dg, db = torch.zeros((M, C), device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = torch.empty_like(x)
    lib = L.lib()
    ws = torch.zeros(lib.mi_bwd_tail_workspace(M, C) // 4 + 64, dtype=torch.float32, device="cuda")
    p = lambda t: t.data_ptr()
    for _ in range(2):
        L.check(lib.mi_bwd_tail(p(dy), M, p(x), C, p(dres), p(mean), p(rstd), p(w), p(gamma), p(beta), p(dx), p(dw), p(dg), p(db), B,
                                H * W, 0, L.MI_BF16, p(ws), torch.cuda.current_stream().cuda_stream), "bwd_tail")
    torch.cuda.synchronize()
    grid = 256 if C == 96 else 512
    mc = M * (C + 1)
    st = ws[: grid * mc].view(grid, mc)[:, : NW * 16].reshape(grid, NW, 16)[:, :, :12].double()
    tiles = B * H * W // 64 / grid
    tot = st.sum(-1)                                  # [grid][NW]
    print(f"C={C} M={M} {H}x{W} bs={B}: {tot.mean().item() / tiles:8.0f} cycles per tile and wave (mean over {grid} workgroups x {NW} waves)")
    for wv in (0, NW - 1):
        sh = st[:, wv, :].mean(0)
        print(f"   wave {wv}: " + ", ".join(f"{n} {100 * sh[i].item() / sh.sum().item():4.1f}%" for i, n in enumerate(NAMES)))
