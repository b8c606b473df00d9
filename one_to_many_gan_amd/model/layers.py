"""Layer modules with the reference's names, constructor signatures and state_dict keys
(src/model/layers.py), executed by the HIP kernels in libo2m_hip.so.

Every module has two entry points:
* ``forward`` -- the reference's public call (logical NCHW in, logical NCHW out);
* ``run``     -- the fused internal call on NHWC buffers used by blocks.py / builder.py,
  where reflection padding, the style modulation/demodulation, bias, activation and
  residual add are folded into the convolution kernel.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import _hip as H
from .. import ops


class EqualisedWeight(nn.Module):
    """Raw N(0,1) parameter + runtime He constant (reference layers.py:12-24)."""

    def __init__(self, shape: list[int]):
        super().__init__()
        self.c = 1 / math.sqrt(math.prod(shape[1:]))
        self.weight = nn.Parameter(torch.randn(shape))

    def forward(self):
        return self.weight * self.c


class EqualisedLinear(nn.Module):
    """Tiny (B x w_dim) linears of the style path: plain torch, they are not on the
    bandwidth- or MFMA-bound part of the step (reference layers.py:27-43)."""

    def __init__(self, in_features: int, out_features: int, bias: float = 0.0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = EqualisedWeight([out_features, in_features])
        self.bias = nn.Parameter(torch.zeros(out_features) + bias)

    def forward(self, x: torch.Tensor):
        return F.linear(x.float(), self.weight(), bias=self.bias)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}"


def _square(kernel_size):
    if isinstance(kernel_size, int):
        return kernel_size, kernel_size
    kh, kw = kernel_size
    return kh, kw


class EqualisedConv2d(nn.Module):
    """Implicit-GEMM conv with equalised-LR weights (reference layers.py:46-108)."""

    def __init__(self, in_features: int, out_features: int, kernel_size, stride: int = 1,
                 padding: int = 0, dilation: int = 1, *, use_bias: bool = True):
        super().__init__()
        if stride != 1 or dilation != 1:
            raise NotImplementedError("the one-to-many GAN only uses stride 1 / dilation 1 convs")
        kh, kw = _square(kernel_size)
        if kh != kw:
            raise NotImplementedError("square kernels only")
        self.in_features, self.out_features = in_features, out_features
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.weight = EqualisedWeight([out_features, in_features, kh, kw])
        self.use_bias = use_bias
        if use_bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        self._prep = None

    def _prepared(self):
        if self._prep is None or self._prep.weight is not self.weight.weight:
            self._prep = ops.PreparedWeight(self.weight.weight, need_q=False)
        return self._prep

    def run(self, t, *, reflect: int = 0, act: int = H.ACT_NONE, residual=None, norm_eps=None, link=None):
        if reflect and self.padding:
            raise ValueError("reflect padding replaces an external ReflectionPad2d: padding must be 0")
        return ops.conv2d(
            t, self.weight.weight, self.bias if self.use_bias else None, self._prepared(),
            pad=reflect or self.padding, pad_mode=H.PAD_REFLECT if reflect else H.PAD_ZERO,
            act=act, residual=residual, norm_eps=norm_eps, link=link)

    def run_norm_act(self, t, *, reflect: int = 0, act: int = H.ACT_NONE, residual=None, eps: float = 1e-5,
                     head_link=None, tail_link=None):
        """conv -> InstanceNorm2d -> activation (+ residual), the statistics coming out of the conv's
        epilogue (builder.py:162-165,170-173,272-282; blocks.py:21-27).  ``head_link``: this conv consumes a
        residual block's input first; ``tail_link``: this norm adds that block's residual (ops.BlockLink)."""
        y, stats = self.run(t, reflect=reflect, norm_eps=eps, link=head_link)
        return ops.instance_norm_act(y, act, residual=residual, eps=eps, stats=stats, link=tail_link)

    def run_norm_act_down(self, t, down, *, act: int = H.ACT_NONE, eps: float = 1e-5):
        """conv -> InstanceNorm2d -> activation -> DownSample ``down`` (builder.py:170-173,272-282) with the last
        three as ONE pass over the conv output when the operator allows it."""
        y, stats = self.run(t, norm_eps=eps)
        if ops.norm_down_fusable(y, down.kind, stats):
            return ops.instance_norm_act_down(y, act, down.kind, stats)
        return down.run(ops.instance_norm_act(y, act, eps=eps, stats=stats))

    def forward(self, x: torch.Tensor):
        return ops.to_public(self.run(ops.to_internal(x)), self.out_features)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features},"
                f" kernel_size={self.kernel_size}, stride={self.stride}, dilation={self.dilation}")


class Conv2dWeightModulate(nn.Module):
    """StyleGAN2 modulated conv (reference layers.py:111-188) computed as ACTIVATION
    modulation: conv(W*c, x*s[b,i]) * rsqrt(sum_i Q[o,i] s[b,i]^2 + eps) -- algebraically
    the reference's per-sample weights, but one dense GEMM with shared weights."""

    def __init__(self, in_features: int, out_features: int, kernel_size: int, w_dim: int,
                 padding: int, *, use_bias: bool = False, demodulate: bool = True, eps: float = 1e-8):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.demodulate, self.padding, self.eps, self.use_bias = demodulate, padding, eps, use_bias
        self.weight = EqualisedWeight([out_features, in_features, kernel_size, kernel_size])
        self.to_style = EqualisedLinear(w_dim, in_features, bias=1)
        if use_bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        self._prep = None

    def _prepared(self):
        if self._prep is None or self._prep.weight is not self.weight.weight:
            self._prep = ops.PreparedWeight(self.weight.weight, need_q=True)
        return self._prep

    def run(self, t, w, *, reflect: int = 0, act: int = H.ACT_NONE, residual=None, link=None):
        if reflect and self.padding:
            raise ValueError("reflect padding replaces an external ReflectionPad2d: padding must be 0")
        return ops.conv2d(
            t, self.weight.weight, self.bias if self.use_bias else None, self._prepared(),
            pad=reflect or self.padding, pad_mode=H.PAD_REFLECT if reflect else H.PAD_ZERO,
            act=act, style=(w, self.to_style.weight.weight, self.to_style.bias), residual=residual,
            demodulate=self.demodulate, eps=self.eps, link=link)

    def forward(self, x: torch.Tensor, w: torch.Tensor):
        return ops.to_public(self.run(ops.to_internal(x), w), self.out_features)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features},"
                f"demodulate={self.demodulate}, padding={self.padding}, eps={self.eps}")


class ReflectFused(nn.Module):
    """ReflectionPad2d(p) followed by a pad-0 conv, with the mirroring done by the conv
    loader (what blocks.py / builder.py use internally); exposed for op-level tests."""

    def __init__(self, conv: nn.Module, p: int):
        super().__init__()
        self.conv, self.p = conv, p

    def forward(self, x, *w):
        return ops.to_public(self.conv.run(ops.to_internal(x), *w, reflect=self.p), self.conv.out_features)


class _Resample(nn.Module):
    kind = "blur"

    def run(self, t):
        return ops.resample(t, self.kind)

    def forward(self, x: torch.Tensor):
        return ops.to_public(self.run(ops.to_internal(x)), x.shape[1])


def _binomial():
    k = torch.tensor([[[[1.0, 2.0, 1.0], [2.0, 4.0, 2.0], [1.0, 2.0, 1.0]]]])
    return k / k.sum()


class Smooth(_Resample):
    """Replicate-pad + 3x3 binomial blur per channel (reference layers.py:191-214).  The
    ``kernel`` buffer is kept for state_dict compatibility; the taps are compiled into the
    banded operator of resample.py."""

    kind = "blur"

    def __init__(self):
        super().__init__()
        self.register_buffer("kernel", _binomial())


class UpSample(_Resample):
    """Bilinear x2 then blur as ONE banded pass (reference layers.py:217-229)."""

    kind = "up"

    def __init__(self):
        super().__init__()
        self.smooth = Smooth()


class DownSample(_Resample):
    """Blur then bilinear to floor(H/2) as ONE banded pass (reference layers.py:232-247)."""

    def __init__(self, *, smooth=True):
        super().__init__()
        self.smooth_map = smooth
        self.smooth = Smooth()
        self.kind = "down" if smooth else "down_nosmooth"
