import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
for (B, Hh, Ci, Co) in [(16, 64, 256, 256), (16, 128, 256, 128), (16, 256, 128, 64)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").to(dt)
    gy = torch.randn(B, Hh, Hh, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    sc = torch.rand(B, Ci, device="cuda")
    fl = 2.0 * B * Hh * Hh * Co * 9 * Ci
    t0 = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=1, pad_mode=H.PAD_REFLECT))
    t1 = timeit(lambda: H.conv2d_wgrad(x, gy, dw, in_scale=sc, pad=1, pad_mode=H.PAD_REFLECT))
    print(f"B{B} {Hh}x{Hh} {Ci}->{Co}: plain {t0*1e6:.0f} us ({fl/t0/1e12:.0f} TF/s)  in_scale {t1*1e6:.0f} us ({fl/t1/1e12:.0f} TF/s)")
