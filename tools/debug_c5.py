"""Where does the tiled bf16 inference output of Restormer base differ from the oracle?  One 256^2 window of a 1024^2 image:
fp32 activations (the parity path), bf16 through the autograd (training) forward, bf16 through the no_grad (fused) forward,
with the fused kernels switched off one by one - all against the CPU oracle."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as m
from image_restoration_amd import ops
from oracle import restormer_ref as R
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_configs import _image

dev = "cuda"
cfg = R.RESTORMER_BASE
sd = R.make_restormer_state(cfg, seed=51)
net = m.Restormer(**cfg); net.load_state_dict(sd); net = net.to(dev).eval()
clean = _image((1, 3, 256, 256), 510)
img = R.degrade_sigma(clean, 25.0, 511).to(torch.bfloat16).float()
t0 = time.time()
with torch.no_grad():
    ref = R.restormer_forward(img, sd, cfg)
print(f"oracle {time.time()-t0:.1f}s; output |max| {float(ref.abs().max()):.3f} rms {float(ref.pow(2).mean().sqrt()):.3f}")

def report(tag, y):
    d = (y.float().cpu() - ref)
    print(f"{tag:40s} max|d|/max|ref| {float(d.abs().max()/ref.abs().max()):.3e}  rms(d)/rms(ref) {float(d.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()):.3e}  "
          f"PSNR(range 1) {10*math.log10(1/float(d.pow(2).mean())):.2f} dB")

x32 = img.to(dev)
x16 = x32.to(torch.bfloat16)
with torch.no_grad():
    report("fp32 no_grad", net(x32))
    report("bf16 no_grad (fused GDFN + fused MDTA)", net(x16))
    for sw in ("MI_NO_FUSED_MDTA", "MI_NO_FUSED_INFER", "MI_NO_LN_HEAD"):
        os.environ[sw] = "1"; m.reload_env()
        report(f"bf16 no_grad, {sw}=1 (cumulative)", net(x16))
    for sw in ("MI_NO_FUSED_MDTA", "MI_NO_FUSED_INFER", "MI_NO_LN_HEAD"):
        os.environ.pop(sw)
    m.reload_env()
y = net(x16.clone().requires_grad_(True))
report("bf16 autograd forward", y.detach())
xb = x16.repeat(25, 1, 1, 1)
with torch.no_grad():
    yb = net(xb)
report("bf16 no_grad, batch of 25 identical (img 0)", yb[:1])
report("bf16 no_grad, batch of 25 identical (img 24)", yb[24:])
