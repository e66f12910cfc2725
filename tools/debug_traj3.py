"""Debug helper: three AdamW steps of Restormer-tiny (fp32) on the GPU against torch's CPU fp32 trajectory: per-tensor metrics."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restormer_ref as R
import image_restoration_amd as m
from image_restoration_amd.trainer import FlatTrainer

cfg = R.RESTORMER_TINY
sd0 = R.make_restormer_state(cfg, seed=2)
clean = torch.from_numpy(np.random.default_rng(77).random((2, 3, 64, 64))).to(torch.float32)
noisy = R.degrade_sigma(clean, 25.0, seed=78)
lr = 1e-3
def oracle(dt):
    ps = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
    opt = torch.optim.AdamW(list(ps.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for _ in range(3):
        opt.zero_grad()
        loss = (R.restormer_forward(noisy.to(dt), ps, cfg) - clean.to(dt)).abs().mean()
        loss.backward()
        opt.step()
    return {k: v.detach().double() for k, v in ps.items()}
p32, p64 = oracle(torch.float32), oracle(torch.float64)
net = m.Restormer(**cfg); net.load_state_dict(sd0); net = net.to("cuda").train()
tr = FlatTrainer(net, lr=lr, weight_decay=0.01)
x, y = noisy.cuda(), clean.cuda()
for step in range(3):
    tr.zero_grad(); loss = (net(x).float() - y).abs().mean(); loss.backward(); tr.reduce_gradients(); tr.optimizer_step()
got = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
tr.close()
def metrics(a, b, tag):
    worst_max = worst_mean = 0.0; worst_cos = 1.0
    for k in sd0:
        w0 = sd0[k].double()
        d = (a[k] - b[k]).abs()
        ua, ub = (a[k] - w0).flatten(), (b[k] - w0).flatten()
        cos = float((ua @ ub) / (ua.norm() * ub.norm()).clamp_min(1e-30))
        worst_max = max(worst_max, float(d.max()) / lr); worst_mean = max(worst_mean, float(d.mean()) / lr); worst_cos = min(worst_cos, cos)
    print(f"{tag}: worst max|d| {worst_max:.3f} lr, worst mean|d| {worst_mean:.4f} lr, worst update cosine {worst_cos:.5f}")
metrics(got, p32, "GPU fp32 vs torch CPU fp32")
metrics(got, p64, "GPU fp32 vs torch CPU fp64")
metrics(p32, p64, "torch CPU fp32 vs torch CPU fp64")
