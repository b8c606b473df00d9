import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dt = torch.bfloat16
B, Hh, Ci, Co = 16, 64, 256, 256
xs = [torch.randn(B, Hh, Hh, Ci, device="cuda").to(dt) for _ in range(5)]
gs = [torch.randn(B, Hh, Hh, Co, device="cuda").to(dt) for _ in range(5)]
dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
fl = 2.0 * 5 * B * Hh * Hh * Co * 9 * Ci
def sep():
    for x, g in zip(xs, gs):
        H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_REFLECT)
t0 = timeit(sep)
print(f"5 separate launches: {t0*1e6:.0f} us ({fl/t0/1e12:.0f} TF/s)")
for sp in (0, 28, 56, 112, 224):
    t1 = timeit(lambda: H.conv2d_wgrad(xs[0], gs[0], dw, pad=1, pad_mode=H.PAD_REFLECT, more=list(zip(xs[1:], gs[1:])), splits=sp))
    print(f"1 launch, 5 segments, splits={sp}: {t1*1e6:.0f} us ({fl/t1/1e12:.0f} TF/s)")
dw2 = torch.zeros_like(dw); dw3 = torch.zeros_like(dw)
for x, g in zip(xs, gs): H.conv2d_wgrad(x, g, dw2, pad=1, pad_mode=H.PAD_REFLECT)
H.conv2d_wgrad(xs[0], gs[0], dw3, pad=1, pad_mode=H.PAD_REFLECT, more=list(zip(xs[1:], gs[1:])))
torch.cuda.synchronize()
print("rel diff", ((dw2 - dw3).norm() / dw2.norm()).item())
