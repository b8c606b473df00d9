"""GPU debugging aid (not a test): raw o2m_conv2d_fwd against torch's own conv on the box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from one_to_many_gan_amd import _hip as H

torch.manual_seed(0)
dev = "cuda"


def run(B, Hh, Ww, Ci, Co, k, pad, dt, ident=False, reflect=False):
    x = torch.randn(B, Hh, Ww, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5
    if ident:
        w.zero_()
        for i in range(min(Ci, Co)):
            w[i, k // 2, k // 2, i] = 1.0
    xd, wd = x.to(dt).contiguous(), w.to(dt).contiguous()
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.full((B, ho, wo, Co), float("nan"), device=dev, dtype=dt)
    H.conv2d_fwd(xd, wd, y, pad=pad, pad_mode=H.PAD_REFLECT if reflect else H.PAD_ZERO, act=H.ACT_NONE)
    torch.cuda.synchronize()
    xin = xd.float().permute(0, 3, 1, 2)
    if reflect:
        xin = F.pad(xin, (pad,) * 4, mode="reflect")
    ref = F.conv2d(xin, wd.float().permute(0, 3, 1, 2), padding=0 if reflect else pad).permute(0, 2, 3, 1)
    err = ((y.float() - ref).norm() / ref.norm()).item()
    nan = torch.isnan(y.float()).sum().item()
    print(f"B{B} {Hh}x{Ww} Ci{Ci} Co{Co} k{k} p{pad} {str(dt)[6:]:8s} ident={ident} refl={reflect}: rel {err:.3e} nan {nan}")
    if err > 1e-2 and ident:
        print(" y[0,0,0,:8]", y[0, 0, 0, :8].float().tolist())
        print(" r[0,0,0,:8]", ref[0, 0, 0, :8].tolist())
        print(" y[0,0,1,:8]", y[0, 0, 1, :8].float().tolist())
        print(" r[0,0,1,:8]", ref[0, 0, 1, :8].tolist())
    return err


for dt in (torch.bfloat16, torch.float32):
    run(1, 8, 8, 8, 8, 1, 0, dt, ident=True)
    run(1, 8, 8, 64, 64, 1, 0, dt, ident=True)
    run(1, 8, 8, 64, 64, 1, 0, dt)
    run(2, 9, 11, 8, 16, 3, 1, dt)
    run(2, 16, 16, 64, 128, 3, 1, dt)
    run(2, 16, 16, 128, 256, 3, 1, dt)
    run(2, 10, 9, 16, 8, 3, 1, dt, reflect=True)
    run(2, 13, 12, 8, 8, 4, 1, dt)
    run(1, 12, 14, 8, 8, 7, 3, dt, reflect=True)
    run(4, 64, 64, 256, 256, 3, 1, dt)
