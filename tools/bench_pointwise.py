"""Micro-benchmark (GPU) of the HBM-bound kernels: achieved GB/s on the hot shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from one_to_many_gan_amd import resample as R
from tools.bench_conv import timeit

dt = torch.bfloat16
dev = "cuda"
SHAPES = [(16, 64, 64, 256), (16, 128, 128, 128), (16, 256, 256, 64), (32, 64, 64, 256), (16, 126, 126, 128)]
for (B, Hh, Ww, C) in SHAPES:
    n = B * Hh * Ww * C
    x = torch.randn(B, Hh, Ww, C, device=dev).to(dt)
    g = torch.randn(B, Hh, Ww, C, device=dev).to(dt)
    y = torch.relu(x)
    out = torch.empty_like(x)
    line = f"B{B} {Hh}x{Ww}x{C}: "
    # act_bwd_reduce: 2 reads + 1 write
    sums = torch.zeros(B, 2, C, device=dev)
    dmul = torch.rand(B, C, device=dev)
    t = timeit(lambda: H.act_bwd_reduce(g, y, None, dmul, out, sums, H.ACT_RELU))
    line += f"actbwd {3*n*2/t/1e9:6.0f} GB/s ({t*1e6:5.0f}us) | "
    # fold_scale_dot pad=1: reads gpad + x, writes gx
    gp = torch.randn(B, Hh + 2, Ww + 2, C, device=dev).to(dt)
    dots = torch.zeros(B, C, device=dev)
    sc = torch.rand(B, C, device=dev)
    t = timeit(lambda: H.fold_scale_dot(gp, x, sc, out, dots, 1))
    line += f"fold {3*n*2/t/1e9:6.0f} ({t*1e6:5.0f}us) | "
    # instnorm stats (1 read), apply (1r+1w), bwd (2r x2 + 1w)
    ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, C), device=dev)
    mr = torch.empty(B, C, 2, device=dev)
    t = timeit(lambda: H.instnorm_stats(x, ws, mr, 1e-5))
    line += f"in_stats {n*2/t/1e9:6.0f} ({t*1e6:5.0f}us) | "
    t = timeit(lambda: H.instnorm_apply(x, mr, None, out, H.ACT_RELU))
    line += f"in_apply {2*n*2/t/1e9:6.0f} ({t*1e6:5.0f}us) | "
    gs = torch.empty(B, C, 2, device=dev)
    t = timeit(lambda: H.instnorm_bwd(g, x, mr, ws, gs, out, H.ACT_RELU))
    line += f"in_bwd {5*n*2/t/1e9:6.0f} ({t*1e6:5.0f}us) | "
    if Hh % 2 == 0:
        for kind in ("up", "down"):
            for tr in (False, True):
                sy, wy, sx, wx, T, ho, wo = R.taps(kind, Hh, Ww, tr, dev) if not tr else R.taps(kind, Hh if kind == "up" else Hh, Ww, tr, dev)
                src = x if not tr else torch.randn(B, (Hh * 2 if kind == "up" else Hh // 2), (Ww * 2 if kind == "up" else Ww // 2), C, device=dev).to(dt)
                dst = torch.empty(B, ho, wo, C, device=dev, dtype=dt)
                t = timeit(lambda: H.resample2d(src, dst, sy, wy, sx, wx, T))
                line += f"{kind}{'T' if tr else ''} {(src.numel()+dst.numel())*2/t/1e9:5.0f} ({t*1e6:4.0f}us T={T}) "
    print(line, flush=True)
