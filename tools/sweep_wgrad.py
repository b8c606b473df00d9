"""Split-K sweep of o2m_conv2d_wgrad: time per shape as a function of the split count.
O2M_WGRAD_TILES=small selects the 128-wide tiles for an A/B against the default tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
SH = [(16, 64, 64, 256, 256, 3, 1), (16, 128, 128, 256, 128, 3, 1), (16, 128, 128, 128, 256, 3, 1),
      (16, 256, 256, 128, 64, 3, 1), (16, 256, 256, 64, 128, 3, 1), (16, 256, 256, 64, 8, 7, 3),
      (16, 256, 256, 8, 64, 7, 3), (32, 64, 64, 256, 256, 3, 1), (32, 62, 62, 128, 256, 4, 1),
      (32, 31, 31, 256, 512, 4, 1)]
SPL = [int(a) for a in sys.argv[1:]] or [0, 8, 12, 16, 20, 28, 42, 56, 84]
for (B, Hh, Ww, Ci, Co, k, pad) in SH:
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
    ho = Hh + 2 * pad - k + 1
    gy = torch.randn(B, ho, ho, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    flops = 2.0 * B * ho * ho * Co * k * k * Ci
    out = []
    for sp in SPL:
        t = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=H.PAD_ZERO, splits=sp), iters=10)
        out.append(f"{'auto' if sp == 0 else sp}:{t*1e6:.0f}us" + (f"({flops/t/1e12:.0f}TF)" if sp == 0 else ""))
    print(f"B{B} {Hh}x{Ww} {Ci}->{Co} k{k}: " + "  ".join(out), flush=True)
