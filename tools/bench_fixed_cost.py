"""Fixed (prologue + epilogue) vs per-K-tile cost of the igemm kernels: same output tile grid, two reduction lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit

def run(B, Hh, Ci, Co, k=3):
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, Hh, Hh, Co, device="cuda", dtype=torch.bfloat16)
    return min(timeit(lambda: H.conv2d_fwd(x, w, y, pad=k // 2, pad_mode=H.PAD_ZERO, act=H.ACT_RELU), iters=20) for _ in range(3))

for (B, Hh, Co, cis) in [(16, 256, 64, (64, 128, 256)), (16, 128, 128, (128, 256, 512)), (16, 64, 256, (128, 256, 512)), (16, 128, 256, (128, 256))]:
    ts = [(ci, run(B, Hh, ci, Co)) for ci in cis]
    (c0, t0), (c1, t1) = ts[0], ts[-1]
    per_kt = (t1 - t0) / ((c1 - c0) * 9 / 64)
    fixed = t0 - per_kt * c0 * 9 / 64
    print(f"B{B} {Hh}x{Hh} Co={Co}: " + "  ".join(f"Ci={c}: {t*1e6:.1f} us" for c, t in ts) +
          f"   -> {per_kt*1e6:.2f} us per K-tile (all rounds), fixed {fixed*1e6:.1f} us", flush=True)
