"""RGB-tail weight gradient: direct (Co 3->8 padded, 7x7) vs space-to-depth (48 outputs, 10x10, stride 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
B, S, Ci = 16, 256, 64
x = torch.randn(B, S, S, Ci, device="cuda").to(dt)
gy = torch.randn(B, S, S, 8, device="cuda").to(dt)
dw = torch.zeros(8, 7, 7, Ci, device="cuda")
t = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=3, pad_mode=H.PAD_REFLECT), iters=10)
print(f"direct 7x7 Co8: {t*1e6:.0f}us")
g2 = torch.randn(B, S // 4, S // 4, 48, device="cuda").to(dt)
dw2 = torch.zeros(48, 10, 10, Ci, device="cuda")
for sp in (0, 4, 8, 12, 16, 24):
    t = timeit(lambda: H.conv2d_wgrad(x, g2, dw2, pad=3, pad_mode=H.PAD_REFLECT, stride=4, splits=sp), iters=10)
    print(f"s2d 10x10 s4 Co48 splits={sp}: {t*1e6:.0f}us")
