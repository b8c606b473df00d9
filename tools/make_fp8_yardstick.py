#!/usr/bin/env python3
"""What does per-tensor fp8 cost the REFERENCE itself?  Runs tests/cases.py on the reference with
``torch.nn.functional.conv2d`` replaced by a fake-quantised convolution that mirrors BASELINE config #5 as
the HIP path implements it (one_to_many_gan_amd/ops.py, set_precision("fp8")):

* forward  : input and filter quantised per tensor to OCP e4m3 (scale = 448 / amax) when the layer has a
             multiple of 128 input channels, exact product otherwise;
* backward : data gradient from the e5m2-quantised output gradient (scale = 57344 / amax) and the e4m3
             filter when the layer has a multiple of 128 output channels; weight gradient from the
             unquantised operands (the HIP path keeps it in bf16).

and compares with the fp32 fixtures (tests/golden/*.npz).  Writes tests/golden/fp8_yardstick.json =
{case: {key: rel_l2}}; the fp8 parity tests accept the HIP path within a small factor of these numbers,
exactly as the bf16 tests do with bf16_yardstick.json.  Build container only (imports /root/reference)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import torch.nn.functional as F
from make_golden import reference_ns
from tests.cases import CASES, run_case

CASES_FP8 = ["conv3_c128", "modconv_c128", "resblock_c128", "gen64_deep"]
_conv2d = F.conv2d


def fq(t, fmt):
    top = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    amax = t.detach().abs().max().clamp_min(1e-12)
    return (t * (top / amax)).clamp(-top, top).to(fmt).to(t.dtype) * (amax / top)


class FakeQuantConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, padding, groups):
        cin_g, cout_g = w.shape[1], w.shape[0] // groups
        ctx.padding, ctx.groups, ctx.q_bwd = padding, groups, cout_g % 128 == 0
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        if cin_g % 128 == 0:
            return _conv2d(fq(x, torch.float8_e4m3fn), fq(w, torch.float8_e4m3fn), bias, 1, padding, 1, groups)
        return _conv2d(x, w, bias, 1, padding, 1, groups)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if ctx.q_bwd:
            gx = torch.nn.grad.conv2d_input(x.shape, fq(w, torch.float8_e4m3fn), fq(g, torch.float8_e5m2),
                                            padding=ctx.padding, groups=ctx.groups)
        else:
            gx = torch.nn.grad.conv2d_input(x.shape, w, g, padding=ctx.padding, groups=ctx.groups)
        gw = torch.nn.grad.conv2d_weight(x, w.shape, g, padding=ctx.padding, groups=ctx.groups)
        gb = g.sum((0, 2, 3)) if ctx.has_bias else None
        return gx, gw, gb, None, None


def fake_conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if w.shape[1] == 1 or stride not in (1, (1, 1)):  # depthwise blur kernels: not a quantised layer
        return _conv2d(x, w, bias, stride, padding, dilation, groups)
    pad = padding if isinstance(padding, int) else padding[0]
    return FakeQuantConv.apply(x, w, bias, pad, groups)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ns = reference_ns()
    out = {}
    F.conv2d = fake_conv2d
    try:
        for name in sys.argv[1:] or CASES_FP8:
            gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
            got = run_case(name, ns, "cpu")
            d = {}
            for k in gold.files:
                if k.endswith("/sum") or k.endswith("/sqsum"):
                    continue
                w = torch.from_numpy(gold[k]).double().flatten()
                d[k] = float((got[k].double().flatten() - w).norm() / (w.norm() + 1e-30)) if w.norm() > 0 else 0.0
            out[name] = d
            worst = sorted(d.items(), key=lambda kv: -kv[1])[:5]
            print(f"{name:16s} max={worst[0][1]:.2e} | " + ", ".join(f"{k}={v:.2e}" for k, v in worst), flush=True)
    finally:
        F.conv2d = _conv2d
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "fp8_yardstick.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
