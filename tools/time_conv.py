"""Time ONE conv shape (HIP events around 20 launches, median of 5 rounds).
Usage: time_conv.py fwd|fp8|wgrad B H W Ci Co k pad reflect [more shapes: 8 numbers each]   (fp8: e4m3 x e4m3 forward)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
mode = sys.argv[1]
nums = [int(v) for v in sys.argv[2:]]
for k0 in range(0, len(nums), 8):
    B, Hh, Ww, Ci, Co, k, pad, refl = nums[k0:k0 + 8]
    dt = torch.bfloat16
    pm = H.PAD_REFLECT if refl else H.PAD_ZERO
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(dt)
    y = torch.empty(B, ho, wo, Co, device="cuda", dtype=dt)
    gy = torch.randn(B, ho, wo, Co, device="cuda").to(dt)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    if mode == "fp8":
        dq = torch.empty(2, 2, device="cuda")
        x8 = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device="cuda"); w8 = torch.empty(w.shape, dtype=torch.float8_e4m3fn, device="cuda")
        H.quantize_fp8(x, x8, dq[0]); H.quantize_fp8(w, w8, dq[1])
    def run():
        if mode == "fwd": H.conv2d_fwd(x, w, y, pad=pad, pad_mode=pm, act=H.ACT_NONE)
        elif mode == "fp8": H.conv2d_fwd(x8, w8, y, pad=pad, pad_mode=pm, act=H.ACT_NONE, deq=dq.view(-1))
        else: H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=pm, p8=True)
    for _ in range(5): run()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    t = sorted(ts)[2]
    fl = 2.0 * B * ho * wo * Co * k * k * Ci
    print(f"{mode} B{B} {Hh}x{Ww} {Ci}->{Co} k{k} pad{pad} refl{refl}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF/s", flush=True)
