"""Diagnostic only: reads the s_memtime phase stamps of the stamped igemm build."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
B, Hh, Ci, Co = 16, 64, 256, 256
dt = torch.bfloat16
x = torch.randn(B, Hh, Hh, Ci, device="cuda").to(dt)
w = (torch.randn(Co, 3, 3, Ci, device="cuda") / 48).to(dt)
y = torch.empty(B, Hh, Hh, Co, device="cuda", dtype=dt)
for _ in range(3):
    H.conv2d_fwd(x, w, y, pad=1, pad_mode=H.PAD_REFLECT, act=H.ACT_RELU)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
H.lib().o2m_debug_stamps.argtypes = [ctypes.c_void_p]
print("rc", H.lib().o2m_debug_stamps(buf))
v = list(buf)
for w_ in range(2):
    a = v[w_ * 4: w_ * 4 + 4]
    tot = sum(a)
    print(f"wave {w_}: per K-stage cycles: dma_issue {a[0]/36:.0f}  compute {a[1]/36:.0f}  vmcnt_wait {a[2]/36:.0f}  barrier {a[3]/36:.0f}  total {tot/36:.0f}")
