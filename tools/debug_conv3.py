import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from image_restoration_amd import ops
torch.manual_seed(0)
B, K, M, H, W = 1, 16, 16, 8, 64
x = torch.randn(B, K, H, W).to(torch.bfloat16)
for tap in range(9):
    w = torch.zeros(M, K, 3, 3)
    for i in range(16):
        w[i, i, tap // 3, tap % 3] = 1.0
    ref = F.conv2d(x.double(), w.double(), padding=1)
    got = ops.conv3x3(x.cuda(), w.cuda()).cpu().double()
    d = (got - ref).abs()
    print("tap", tap, "max err", float(d.max()), "bad rows", sorted(set(torch.nonzero(d > 1e-3)[:, 2].tolist()))[:12],
          "bad cols", sorted(set(torch.nonzero(d > 1e-3)[:, 3].tolist()))[:20], "bad ch", sorted(set(torch.nonzero(d > 1e-3)[:, 1].tolist()))[:16])
# channel mixing: w[m][k] = 1 at centre tap for one (m,k)
for (m, k) in ((0, 5), (7, 0), (15, 15), (3, 9)):
    w = torch.zeros(M, K, 3, 3); w[m, k, 1, 1] = 1.0
    got = ops.conv3x3(x.cuda(), w.cuda()).cpu().double()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    nzc = sorted(set(torch.nonzero(got.abs() > 1e-6)[:, 1].tolist()))
    print("m,k", m, k, "err", float((got - ref).abs().max()), "nonzero out channels", nzc)
