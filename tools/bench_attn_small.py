"""The c x c side of MDTA alone (attn_fold forward; attn_bwd_partial + attn_bwd_finish backward) at the Restormer-base shapes,
read from the library's per-kernel event table.  Run on the GPU box: python tools/bench_attn_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_restoration_amd import ops  # noqa: E402

B = int(os.environ.get("BA_BATCH", "32"))
for C, heads, H, W in ((48, 1, 64, 64), (96, 2, 64, 64), (96, 1, 64, 64), (192, 4, 64, 64), (384, 8, 32, 32)):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16)
    att = (torch.ones(heads, 1, 1, device="cuda"), torch.randn(3 * C, C, 1, 1, device="cuda") / C ** 0.5, None,
           torch.randn(3 * C, 1, 3, 3, device="cuda") / 3, None, torch.randn(C, C, 1, 1, device="cuda") / C ** 0.5, None)
    p = [t.requires_grad_(True) if t is not None else None for t in att]
    from image_restoration_amd import torch_ops
    xr = x.clone().requires_grad_(True)

    def step():
        out = torch_ops.mdta(xr, heads, p)
        out.float().sum().backward()

    for _ in range(3):
        step()
    ops.prof_enable(True)
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    tab = ops.prof_collect()
    ops.prof_enable(False)
    line = ", ".join(f"{k} {tab[k]['ms'] / tab[k]['launches'] * 1e3:6.1f} us" for k in ("attn_fold", "attn_bwd_small", "gram_reduce") if k in tab)
    print(f"C={C:3d} heads={heads} c={C // heads:3d} bs={B}: {line}", flush=True)
