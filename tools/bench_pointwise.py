"""Pointwise / reduction kernels at the step's main shapes: time and achieved HBM GB/s
(algorithmic bytes: every operand read once, every result written once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from one_to_many_gan_amd import resample as R
from tools.bench_conv import timeit

dt = torch.bfloat16
SHAPES = [(16, 64, 64, 256), (16, 128, 128, 128), (16, 256, 256, 64), (16, 128, 128, 256), (16, 256, 256, 128)]
only = sys.argv[1] if len(sys.argv) > 1 else ""


def rnd(*s):
    return torch.randn(*s, device="cuda").to(dt)


def show(name, shape, t, nbytes):
    print(f"{name:18s} {str(shape):22s} {t*1e6:8.1f} us {nbytes/t/1e12:6.2f} TB/s", flush=True)


for (B, Hh, Ww, C) in SHAPES:
    n = B * Hh * Ww * C
    e = 2 * n  # bytes of one bf16 tensor
    g, y, x, res = rnd(B, Hh, Ww, C), rnd(B, Hh, Ww, C).relu_(), rnd(B, Hh, Ww, C), rnd(B, Hh, Ww, C)
    out = torch.empty_like(g)
    out2 = torch.empty_like(g)
    d = torch.rand(B, C, device="cuda") + 0.5
    if only in ("", "act"):
        sums = torch.zeros(B, 2, C, device="cuda")
        t = timeit(lambda: H.act_bwd_reduce(g, y, None, d, out, sums, H.ACT_RELU))
        show("act_bwd_reduce", (B, Hh, Ww, C), t, 3 * e)
        t = timeit(lambda: H.act_bwd_reduce(g, None, None, None, None, sums, H.ACT_NONE))
        show("reduce_only", (B, Hh, Ww, C), t, e)
    if only in ("", "fold"):
        dots = torch.zeros(B, C, device="cuda")
        gp = rnd(B, Hh + 2, Ww + 2, C)
        t = timeit(lambda: H.fold_scale_dot(gp, x, d, out, dots, 1, xs=out2))
        show("fold_scale_dot", (B, Hh, Ww, C), t, 4 * e)
    if only in ("", "in"):
        ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, C), device="cuda")
        mr = torch.empty(B, C, 2, device="cuda")
        t = timeit(lambda: H.instnorm_stats(x, ws, mr, 1e-5))
        show("in_stats", (B, Hh, Ww, C), t, e)
        t = timeit(lambda: H.instnorm_apply(x, mr, None, out, H.ACT_RELU))
        show("in_apply", (B, Hh, Ww, C), t, 2 * e)
        gs = torch.empty(B, C, 2, device="cuda")
        t = timeit(lambda: H.instnorm_bwd(g, x, mr, ws, gs, out, H.ACT_RELU))
        show("in_bwd(2 pass)", (B, Hh, Ww, C), t, 5 * e)
    if only in ("", "res"):
        for kind, tr in (("down", False), ("up", False), ("blur", False), ("up", True), ("down", True)):
            if kind == "up" and not tr and Hh > 128:
                continue
            sy, wy, sx, wx, T, ho, wo = R.taps(kind, Hh, Ww, tr, "cuda")
            if tr:
                kind += "^T"
            from one_to_many_gan_amd import ops
            tp = (sy, wy, sx, wx, T, ho, wo)
            yy = ops._apply_taps(x, tp)
            t = timeit(lambda: ops._apply_taps(x, tp))
            show("resample_" + kind, (B, Hh, Ww, C), t, e + 2 * yy.numel())
