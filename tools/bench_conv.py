"""Kernel micro-benchmark (GPU): igemm forward / wgrad on the hot shapes, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H

dev = "cuda"
SHAPES = [  # B, H, W, Ci, Co, k, pad, reflect
    (16, 64, 64, 256, 256, 3, 1, True),
    (32, 64, 64, 256, 256, 3, 1, True),
    (16, 128, 128, 256, 128, 3, 1, False),
    (16, 256, 256, 128, 64, 3, 1, False),
    (16, 256, 256, 64, 128, 3, 1, False),
    (16, 128, 128, 128, 256, 3, 1, False),
    (16, 256, 256, 64, 8, 7, 3, True),
    (16, 256, 256, 8, 64, 7, 3, True),
    (16, 127, 127, 64, 128, 4, 1, False),
    (16, 63, 63, 128, 256, 4, 1, False),
    (16, 31, 31, 256, 512, 4, 1, False),
    # round 3: the batched decoder groups (3B = 48, 2B = 32) and the encoder's 2B
    (48, 64, 64, 256, 256, 3, 1, True),
    (48, 128, 128, 256, 128, 3, 1, False),
    (48, 128, 128, 128, 256, 3, 1, False),
    (48, 256, 256, 128, 64, 3, 1, False),
    (48, 256, 256, 64, 128, 3, 1, False),
    (32, 256, 256, 64, 128, 3, 1, False),
    (32, 128, 128, 128, 256, 3, 1, False),
    (32, 127, 127, 64, 128, 4, 1, False),
    (32, 63, 63, 128, 256, 4, 1, False),
    (32, 31, 31, 256, 512, 4, 1, False),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
  dt = torch.bfloat16
  which = sys.argv[1] if len(sys.argv) > 1 else "all"
  for (B, Hh, Ww, Ci, Co, k, pad, refl) in SHAPES:
      x = torch.randn(B, Hh, Ww, Ci, device=dev).to(dt)
      w = (torch.randn(Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5).to(dt)
      ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
      y = torch.empty(B, ho, wo, Co, device=dev, dtype=dt)
      pm = H.PAD_REFLECT if refl else H.PAD_ZERO
      flops = 2.0 * B * ho * wo * Co * k * k * Ci
      line = f"B{B} {Hh}x{Ww} {Ci}->{Co} k{k}: "
      if which in ("all", "fwd"):
          t = timeit(lambda: H.conv2d_fwd(x, w, y, pad=pad, pad_mode=pm, act=H.ACT_RELU))
          line += f"fwd {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s | "
      if which in ("all", "wgrad"):
          gy = torch.randn(B, ho, wo, Co, device=dev).to(dt)
          dw = torch.zeros(Co, k, k, Ci, device=dev)
          t = timeit(lambda: H.conv2d_wgrad(x, gy, dw, pad=pad, pad_mode=pm, p8=os.environ.get('O2M_WGRAD_P8', '1') == '1'))
          line += f"wgrad {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s"
      print(line, flush=True)


if __name__ == "__main__":
    main()
