"""Where does the host time of a MoCE-IR training step go?  (cProfile over a few eager steps; run on the GPU box.)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_restoration_amd import configs  # noqa: E402
from image_restoration_amd.moce_ir import MoCEIR  # noqa: E402
from image_restoration_amd.trainer import FlatTrainer  # noqa: E402
from image_restoration_amd import ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
model = MoCEIR(**configs.MOCEIR_BASE).to(dev).train()
tr = FlatTrainer(model, lr=2e-4)
x = torch.rand((8, 3, 128, 128), device=dev).to(torch.bfloat16)
y = torch.rand((8, 3, 128, 128), device=dev).to(torch.bfloat16)


def step():
    tr.zero_grad()
    out = model(x)
    loss, dout = ops.l1_loss(out, y, want_grad=True)
    aux = model.total_loss
    torch.autograd.backward([out, aux], [dout, torch.full_like(aux, 0.01)])
    tr.reduce_gradients()
    tr.optimizer_step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(45)
