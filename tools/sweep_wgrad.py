import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
SH = [(16, 64, 64, 256, 256, 3, 1), (16, 128, 128, 256, 128, 3, 1), (16, 256, 256, 128, 64, 3, 1),
      (16, 256, 256, 64, 128, 3, 1), (16, 256, 256, 64, 8, 7, 3), (16, 256, 256, 8, 64, 7, 3), (32, 64, 64, 256, 256, 3, 1)]
for (B, Hh, Ww, Ci, Co, k, pad) in SH:
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
    ho = Hh + 2 * pad - k + 1
    gy = torch.randn(B, ho, ho, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    flops = 2.0 * B * ho * ho * Co * k * k * Ci
    tiles = -(-Co // (128 if Co > 64 else (64 if Co > 32 else 32))) * -(-(k * k * Ci) // 128)
    out = []
    for blocks in (0, 512, 768, 1024, 1536, 2048, 3072, 4096):
        sp = 0 if blocks == 0 else max(1, blocks // tiles)
        t = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=H.PAD_ZERO, splits=sp), iters=10)
        out.append(f"{'auto' if blocks == 0 else blocks}:{t*1e6:.0f}us")
    print(f"B{B} {Hh}x{Ww} {Ci}->{Co} k{k} tiles={tiles}: " + "  ".join(out), flush=True)
