"""Debug helper: high-water mark of the deferred-reduction arena over one Restormer-base bs-32 step (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI_DEFER_MB", "16384")
import image_restoration_amd as m
from image_restoration_amd import _lib
from image_restoration_amd.configs import RESTORMER_BASE as cfg
from image_restoration_amd.trainer import FlatTrainer
net = m.Restormer(**cfg).to("cuda").to(torch.bfloat16).train() if False else m.Restormer(**cfg).to("cuda").train()
tr = FlatTrainer(net, lr=1e-4)
x = torch.rand(int(os.environ.get("HW_BATCH", "32")), 3, 256, 256, device="cuda").bfloat16()
for _ in range(2):
    tr.zero_grad(); (net(x).float() - x.float()).abs().mean().backward(); tr.reduce_gradients(); tr.optimizer_step()
torch.cuda.synchronize()
print("deferred arena high water: %.1f MB" % (_lib.lib().mi_deferred_high_water() / 2**20))
tr.close()
