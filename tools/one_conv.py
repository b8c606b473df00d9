"""Run ONE conv shape a few times (for rocprofv3 --pmc runs).
Usage: one_conv.py fwd|wgrad [B H W Ci Co k pad reflect]   (default: the 3x3 256->256 64x64 B=16 layer)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, Hh, Ww, Ci, Co, k, pad, refl = ([int(v) for v in sys.argv[2:10]] if len(sys.argv) >= 10 else [16, 64, 64, 256, 256, 3, 1, 1])
dt = torch.bfloat16
pm = H.PAD_REFLECT if refl else H.PAD_ZERO
ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(dt)
y = torch.empty(B, ho, wo, Co, device="cuda", dtype=dt)
gy = torch.randn(B, ho, wo, Co, device="cuda").to(dt)
dw = torch.zeros(Co, k, k, Ci, device="cuda")
for _ in range(5):
    if mode == "fwd":
        H.conv2d_fwd(x, w, y, pad=pad, pad_mode=pm, act=H.ACT_RELU)
    else:
        H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=pm, p8=True)
torch.cuda.synchronize()
