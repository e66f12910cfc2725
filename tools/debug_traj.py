"""Debug helper: per-parameter gradient error of ONE Restormer-tiny fp32 step against the fp64 oracle (GPU box)."""
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
ps = {k: v.clone().double().requires_grad_(True) for k, v in sd0.items()}
loss = (R.restormer_forward(noisy.double(), ps, cfg) - clean.double()).abs().mean()
loss.backward()
net = m.Restormer(**cfg)
net.load_state_dict(sd0)
net = net.to("cuda").train()
use_tr = os.environ.get("DT_TRAINER", "1") == "1"
tr = FlatTrainer(net, lr=1e-3, weight_decay=0.01) if use_tr else None
if tr: tr.zero_grad()
l2 = (net(noisy.cuda()).float() - clean.cuda()).abs().mean()
l2.backward()
if tr: tr.reduce_gradients()
print("loss", float(loss), float(l2))
rows = []
for k, p in net.named_parameters():
    g = (p.main_grad if (tr and hasattr(p, "main_grad")) else p.grad).detach().double().cpu()
    r = ps[k].grad
    rows.append((float((g - r).abs().max() / r.abs().max().clamp_min(1e-30)), k, float(r.abs().max())))
rows.sort(reverse=True)
for e, k, mx in rows[:12]:
    print(f"{e:10.3e}  {k:50s} max|ref| {mx:.3e}")
if tr: tr.close()
