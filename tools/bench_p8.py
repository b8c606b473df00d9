"""A/B of the igemm variants on the 256x256-tile layers (run under O2M_IGEMM_P8=0|1|2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
SHAPES = [(16, 64, 64, 256, 256, 3, 1, True), (32, 64, 64, 256, 256, 3, 1, True), (16, 66, 66, 256, 256, 3, 2, False),
          (16, 128, 128, 128, 256, 3, 1, False), (8, 64, 64, 512, 512, 3, 1, True)]
for (B, Hh, Ww, Ci, Co, k, pad, refl) in SHAPES:
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.empty(B, ho, wo, Co, device="cuda", dtype=torch.bfloat16)
    pm = H.PAD_REFLECT if refl else H.PAD_ZERO
    flops = 2.0 * B * ho * wo * Co * k * k * Ci
    ts = [timeit(lambda: H.conv2d_fwd(x, w, y, pad=pad, pad_mode=pm, act=H.ACT_RELU), iters=30) for _ in range(3)]
    t = min(ts)
    print(f"P8={os.environ.get('O2M_IGEMM_P8','1')} B{B} {Hh}x{Ww} {Ci}->{Co} pad{pad}: {t*1e6:7.1f} us {flops/t/1e12:7.1f} TF/s", flush=True)

# fp8 form (config #5) on the same layers: e4m3 x e4m3, quantisation outside the timed region
for (B, Hh, Ww, Ci, Co, k, pad, refl) in SHAPES:
    if Ci % 128:
        continue
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    dq = torch.empty(2, 2, device="cuda")
    x8, w8 = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device="cuda"), torch.empty(w.shape, dtype=torch.float8_e4m3fn, device="cuda")
    H.quantize_fp8(x, x8, dq[0])
    H.quantize_fp8(w, w8, dq[1])
    deq = dq.view(-1)  # {1/scale_x, amax_x, 1/scale_w, amax_w}
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.empty(B, ho, wo, Co, device="cuda", dtype=torch.bfloat16)
    pm = H.PAD_REFLECT if refl else H.PAD_ZERO
    flops = 2.0 * B * ho * wo * Co * k * k * Ci
    t = min(timeit(lambda: H.conv2d_fwd(x8, w8, y, pad=pad, pad_mode=pm, act=H.ACT_RELU, deq=deq), iters=30) for _ in range(3))
    tq = min(timeit(lambda: H.quantize_fp8(x, x8, dq[0]), iters=30) for _ in range(2))
    print(f"fp8  B{B} {Hh}x{Ww} {Ci}->{Co} pad{pad}: {t*1e6:7.1f} us {flops/t/1e12:7.1f} TF/s   (quantising x: {tq*1e6:.1f} us)", flush=True)
