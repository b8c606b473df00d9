import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
for (Hh, Ci, Co) in [(64, 256, 256), (128, 128, 128), (128, 256, 128), (256, 128, 64)]:
    out = []
    for B in (8, 16, 32):
        x = torch.randn(B, Hh, Hh, Ci, device="cuda").to(torch.bfloat16)
        g = torch.randn(B, Hh, Hh, Co, device="cuda").to(torch.bfloat16)
        dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        t = min(timeit(lambda: H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_ZERO), iters=20) for _ in range(3))
        out.append((B, t))
    fl = lambda B: 2.0 * B * Hh * Hh * Co * 9 * Ci
    print(f"{Hh}x{Hh} {Ci}->{Co}: " + "  ".join(f"B={B}: {t*1e6:.1f} us ({fl(B)/t/1e12:.0f} TF/s)" for B, t in out), flush=True)
