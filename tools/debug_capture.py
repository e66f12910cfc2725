"""Which part of a MoCE-IR step breaks HIP-graph capture?  Each case runs in a child process (a failed capture can take the
process down).  python tools/debug_capture.py"""
import os
import subprocess
import sys

CASES = ["restormer_tiny_step", "moce_fwd", "moce_fwd_bwd", "moce_step", "adapter_fwd_bwd", "encoder_fwd_bwd", "route_only", "fft_fwd_bwd"]

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
os.environ["MI_DEFER_MB"] = "0"
import image_restoration_amd as m
from image_restoration_amd import configs, ops, moce_ir
from image_restoration_amd.trainer import FlatTrainer
case = sys.argv[1]
dev = "cuda"
torch.manual_seed(0)
def capture(fn, warm=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    g.replay(); g.replay()
    torch.cuda.synchronize()
if case == "restormer_tiny_step":
    net = m.Restormer(**configs.RESTORMER_TINY).to(dev)
    tr = FlatTrainer(net, lr=1e-4)
    x = torch.rand(2, 3, 64, 64, device=dev).to(torch.bfloat16)
    def fn():
        tr.zero_grad(); net(x).float().abs().mean().backward(); tr.reduce_gradients(); tr.optimizer_step(use_dev_scalars=True)
    tr.set_step_scalars(1)
    capture(fn)
elif case in ("moce_fwd", "moce_fwd_bwd", "moce_step"):
    net = moce_ir.MoCEIR(**configs.MOCEIR_BASE).to(dev).train()
    tr = FlatTrainer(net, lr=1e-4)
    x = torch.rand(4, 3, 128, 128, device=dev).to(torch.bfloat16)
    def fn():
        tr.zero_grad()
        if case == "moce_fwd":
            with torch.no_grad():
                net(x)
            return
        out = net(x)
        loss = out.float().abs().mean() + 0.01 * net.total_loss
        loss.backward()
        if case == "moce_step":
            tr.reduce_gradients(); tr.optimizer_step(use_dev_scalars=True)
    tr.set_step_scalars(1)
    capture(fn)
elif case in ("adapter_fwd_bwd", "route_only"):
    ad = moce_ir.AdapterLayer(48, rank=4, num_experts=4, top_k=1, expert_layer=moce_ir.FFTAttention, stage_depth=1,
                              depth_type="constant", rank_type="spread", freq_dim=32, with_complexity=True).to(dev).train()
    x = torch.randn(4, 48, 32, 32, device=dev).to(torch.bfloat16); sh = torch.randn_like(x); fe = torch.randn(4, 32, device=dev)
    def fn():
        for p in ad.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        if case == "route_only":
            out = ad.routing.route(xx, fe)
            (out[0].sum() + out[3].sum()).backward()
            return
        y = ad(xx, fe, sh)
        (y.float().square().mean() + 0.1 * ad.loss).backward()
    capture(fn)
elif case in ("gap_fwd_bwd", "route_bwd_call", "route_fwd_call", "route_gates_only", "route_aux_only", "route_nograd_params"):
    ad = moce_ir.AdapterLayer(48, rank=4, num_experts=4, top_k=1, expert_layer=moce_ir.FFTAttention, stage_depth=1,
                              depth_type="constant", rank_type="spread", freq_dim=32, with_complexity=True).to(dev).train()
    rt = ad.routing
    x = torch.randn(4, 48, 32, 32, device=dev).to(torch.bfloat16); fe = torch.randn(4, 32, device=dev)
    pooled = torch.randn(4, 48, device=dev); noise = torch.randn(4, 4, device=dev)
    cx = rt.complexity.float()
    if case == "route_nograd_params":
        for p in rt.parameters():
            p.requires_grad_(False)
    def fn():
        for p in ad.parameters():
            p.grad = None
        if case == "gap_fwd_bwd":
            xx = x.clone().requires_grad_(True)
            moce_ir._GapFn.apply(xx).sum().backward()
        elif case == "route_fwd_call":
            ops.moe_route_fwd(pooled, fe, rt.gate[2].weight, rt.freq_gate.weight, noise, cx, 1, True)
        elif case == "route_bwd_call":
            gates, idx, vals, aux, tb = ops.moe_route_fwd(pooled, fe, rt.gate[2].weight, rt.freq_gate.weight, noise, cx, 1, True)
            ops.moe_route_bwd(pooled, fe, rt.gate[2].weight, rt.freq_gate.weight, noise, cx, tb, idx, torch.ones_like(gates), None,
                              torch.ones_like(aux), True)
        else:
            xx = x.clone().requires_grad_(True)
            out = rt.route(xx, fe)
            loss = out[0].sum() if case in ("route_gates_only", "route_nograd_params") else out[3].sum()
            loss.backward()
    capture(fn)
elif case == "encoder_fwd_bwd":
    blk = moce_ir.EncoderBlock(48, 1, 2.66, False, "WithBias").to(dev)
    x = torch.randn(2, 48, 32, 64, device=dev).to(torch.bfloat16)
    def fn():
        for p in blk.parameters():
            p.grad = None
        blk(x.clone().requires_grad_(True)).float().sum().backward()
    capture(fn)
elif case == "fft_fwd_bwd":
    f = moce_ir.FFTAttention(24, kernel_size=3, patch_size=8).to(dev)
    x = torch.randn(4, 24, 32, 32, device=dev).to(torch.bfloat16)
    def fn():
        for p in f.parameters():
            p.grad = None
        f(x.clone().requires_grad_(True)).float().sum().backward()
    capture(fn)
print("CAPTURE OK", case)
'''

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for c in (sys.argv[1:] or CASES):
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, c], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=300)
        tail = [l for l in r.stdout.splitlines() if "amdgpu.ids" not in l][-6:]
        print(f"== {c}: rc {r.returncode}\n   " + "\n   ".join(t[:220] for t in tail), flush=True)
