#!/usr/bin/env python3
"""Do two HBM-bound kernels of this library overlap usefully on two HIP streams?  (gram 510x96 and pw_gemm 96<-510 at bs 32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd import ops

B, H, W = 32, 256, 256
dy = torch.randn(B, 510, H, W, device="cuda").bfloat16()
x = torch.randn(B, 96, H, W, device="cuda").bfloat16()
w = torch.randn(510, 96, device="cuda")
s2 = torch.cuda.Stream()

def seq():
    ops.gram(dy, x, 1, True)
    ops.conv1x1(dy, w, None, None, True)

def par():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        ops.gram(dy, x, 1, True)
    ops.conv1x1(dy, w, None, None, True)
    torch.cuda.current_stream().wait_stream(s2)

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:12s}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per pair")
