import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
for (B, Hh, Ww, Ci, Co, k, pad) in [(16, 64, 64, 256, 256, 3, 1), (16, 128, 128, 256, 128, 3, 1), (16, 256, 256, 128, 64, 3, 1), (16, 31, 31, 256, 512, 4, 1)]:
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
    ho = Hh + 2 * pad - k + 1
    gy = torch.randn(B, ho, ho, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    flops = 2.0 * B * ho * ho * Co * k * k * Ci
    out = []
    for sp in (0, 4, 8, 12, 16, 24, 32, 48, 64):
        t = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=H.PAD_ZERO, splits=sp))
        out.append(f"{sp}:{flops/t/1e12:.0f}")
    print(f"B{B} {Hh}x{Ww} {Ci}->{Co} k{k}: " + "  ".join(out), flush=True)
