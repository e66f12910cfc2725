"""Does running a producer -> consumer kernel chain in batch chunks that fit the 256 MiB Infinity Cache beat running each kernel
over the whole batch?  GDFN forward chain (project_in -> dw3x3 + gate -> project_out) and MDTA forward chain at C = 96, 256^2,
bs 32: whole batch per kernel vs chunks of 1 / 2 / 4 / 8 images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as m
from image_restoration_amd import ops
from oracle import restormer_ref as R

dev = "cuda"
B = int(os.environ.get("BC_BATCH", "32"))


def run(c, heads, hw, chunks):
    sd = R.make_block_state(c, heads, 2.66, False, "WithBias", seed=1)
    P = lambda k: sd[k].to(dev).float().contiguous()
    ffn = (P("ffn.project_in.weight"), None, P("ffn.dwconv.weight"), None, P("ffn.project_out.weight"), None)
    att = (P("attn.temperature"), P("attn.qkv.weight"), None, P("attn.qkv_dwconv.weight"), None, P("attn.project_out.weight"), None)
    x = torch.randn((B, c) + hw, device=dev).to(torch.bfloat16)
    res = {}
    for name, fn in (("gdfn_fwd(saved)", lambda xx: ops.gdfn_fwd(xx, xx, ffn, True)), ("mdta_fwd(saved)", lambda xx: ops.mdta_fwd(xx, xx, att, heads, True))):
        for n in chunks:
            def go():
                outs = []
                for b0 in range(0, B, n):
                    outs.append(fn(x[b0:b0 + n]))
                return outs
            go(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                keep = go()
            torch.cuda.synchronize()
            res[(name, n)] = (time.perf_counter() - t0) / 5 * 1e3
            del keep
    return res


for c, heads, hw in ((96, 1, (256, 256)), (48, 1, (256, 256)), (96, 2, (128, 128))):
    r = run(c, heads, hw, [B, 8, 4, 2, 1])
    for (name, n), ms in r.items():
        print(f"C={c} {hw[0]}^2 bs{B} {name:18s} chunk {n:3d}: {ms:8.3f} ms")
