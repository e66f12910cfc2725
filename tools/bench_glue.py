#!/usr/bin/env python3
"""How much of a training step is U-Net glue (patch embed, down/up-sampling, skip reductions, output conv)?
Times Restormer base fwd+bwd at bs 8 / 256^2 / bf16 with every TransformerBlock replaced by identity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as m
from oracle import restormer_ref as R

dev = torch.device("cuda")
torch.manual_seed(0)
net = m.Restormer(**R.RESTORMER_BASE).to(dev)
for name in ("encoder_level1", "encoder_level2", "encoder_level3", "latent", "decoder_level3", "decoder_level2",
             "decoder_level1", "refinement"):
    setattr(net, name, torch.nn.Identity())
import os
x = torch.rand(int(os.environ.get("BK_BATCH", "8")), 3, 256, 256, device=dev).bfloat16()
def step():
    for p in net.parameters():
        p.grad = None
    y = net(x)
    y.backward(torch.ones_like(y))
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"glue-only fwd+bwd: {(time.perf_counter() - t0) * 100:.2f} ms/step (eager)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print(f"glue-only fwd+bwd: {(time.perf_counter() - t0) * 100:.2f} ms/step (graph)")
