"""Residual blocks with the reference's names and state_dict keys (src/model/blocks.py).

The ``conv_block`` containers keep the reference's slot layout (so ``conv_block.1`` /
``conv_block.5`` / ``conv_block.4`` carry the parameters), but the block is executed as a
fixed fused plan: reflection pads live in the conv loaders, ReLU and the residual add in
the conv / norm epilogues.
"""

from __future__ import annotations

import torch
from torch import nn

from .. import _hip as H
from .. import ops
from .layers import Conv2dWeightModulate, EqualisedConv2d


class _Slot(nn.Module):
    """Parameter-free position in a reference ``Sequential`` (pad / norm / activation); the
    work it stands for is fused into the neighbouring kernel."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return f"fused: {self.what}"


class ResnetBlock(nn.Module):
    """x + IN(conv(rpad(ReLU(IN(conv(rpad(x))))))) (reference blocks.py:9-33)."""

    def __init__(self, dim: int, *, use_bias: bool = False):
        super().__init__()
        self.dim = dim
        def conv3():
            return EqualisedConv2d(dim, dim, 3, padding=0, use_bias=use_bias)

        # slots 0..6 as in the reference Sequential: pad, CONV, norm, relu, pad, CONV, norm
        self.conv_block = nn.Sequential(
            _Slot("ReflectionPad2d(1) -> conv loader"), conv3(), _Slot("InstanceNorm2d -> instnorm kernels"),
            _Slot("ReLU -> instnorm apply"), _Slot("ReflectionPad2d(1) -> conv loader"), conv3(),
            _Slot("InstanceNorm2d (+ residual add) -> instnorm kernels"))

    def run(self, t):
        link = ops.BlockLink()  # the residual's gradient is added by the first conv's fold kernel
        u = self.conv_block[1].run_norm_act(t, reflect=1, act=H.ACT_RELU, head_link=link)
        return self.conv_block[5].run_norm_act(u, reflect=1, act=H.ACT_NONE, residual=t, tail_link=link)

    def forward(self, x: torch.Tensor):
        return ops.to_public(self.run(ops.to_internal(x)), self.dim)


class ModulatedResnetBlock(nn.Module):
    """x + modconv(rpad(ReLU(modconv(rpad(x), w))), w) (reference blocks.py:36-68); both
    convs get the same ``w`` but own their ``to_style``."""

    def __init__(self, dim: int, w_dim: int, *, use_bias: bool = False):
        super().__init__()
        self.dim = dim
        def modconv3():
            return Conv2dWeightModulate(dim, dim, 3, w_dim, 0, use_bias=use_bias)

        # slots 0..4 as in the reference ModuleList: pad, MODCONV, relu, pad, MODCONV
        self.conv_block = nn.ModuleList([
            _Slot("ReflectionPad2d(1) -> conv loader"), modconv3(), _Slot("ReLU -> conv epilogue"),
            _Slot("ReflectionPad2d(1) -> conv loader"), modconv3()])

    def run(self, t, w):
        # u is consumed by the second conv only and t reaches its residual unchanged: the backward of the two
        # convs is chained through a BlockLink (residual gradient added by the first conv's fold kernel; the
        # second conv's fold fused with the first conv's ReLU backward)
        link = ops.BlockLink(fuse_act=True)
        u = self.conv_block[1].run(t, w, reflect=1, act=H.ACT_RELU, link=link)
        return self.conv_block[4].run(u, w, reflect=1, residual=t, link=link)

    def forward(self, x: torch.Tensor, w: torch.Tensor):
        return ops.to_public(self.run(ops.to_internal(x), w), self.dim)
