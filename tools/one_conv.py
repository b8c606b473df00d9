"""Run ONE conv shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
B, Hh, Ww, Ci, Co, k, pad = 16, 64, 64, 256, 256, 3, 1
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dt = torch.bfloat16
x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
w = (torch.randn(Co, k, k, Ci, device="cuda") / 48).to(dt)
y = torch.empty(B, Hh, Ww, Co, device="cuda", dtype=dt)
gy = torch.randn(B, Hh, Ww, Co, device="cuda").to(dt)
dw = torch.zeros(Co, k, k, Ci, device="cuda")
for _ in range(5):
    if mode == "fwd":
        H.conv2d_fwd(x, w, y, pad=pad, pad_mode=H.PAD_REFLECT, act=H.ACT_RELU)
    else:
        H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=H.PAD_REFLECT)
torch.cuda.synchronize()
