"""Fixed cost and per-K-tile cost of the p8 igemm kernel: time the same 256 -> 256 layer (64 x 64 maps) with 1x1, 3x3, 5x5
and 7x7 zero-padded filters (4 / 36 / 100 / 196 K-tiles) at one, two and three rounds of tiles; least-squares line."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from bench_conv import timeit

dt = torch.bfloat16
for B in (16, 32, 48):
    pts = []
    for k in (1, 3, 5, 7):
        x = torch.randn(B, 64, 64, 256, device="cuda").to(dt)
        w = (torch.randn(256, k, k, 256, device="cuda") / (256 * k * k) ** 0.5).to(dt)
        y = torch.empty(B, 64, 64, 256, device="cuda", dtype=dt)
        t = timeit(lambda: H.conv2d_fwd(x, w, y, pad=k // 2, pad_mode=H.PAD_ZERO, act=H.ACT_RELU), iters=30)
        pts.append((k * k * 4, t * 1e6))
    n = len(pts)
    sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
    rounds = B // 16
    print(f"B={B}: " + "  ".join(f"{kt} K-tiles {t:7.1f} us" for kt, t in pts) +
          f"  | per K-tile and round {slope / rounds:.3f} us ({256*256*64*2*256/ (slope / rounds) / 1e6:.0f} TF/s in-loop), fixed {icpt:.1f} us per launch", flush=True)
