"""The four networks with the reference's public API and state_dict keys
(src/model/builder.py), executed as fused plans over the HIP kernels.

Public tensors are logical NCHW; inside a network everything flows as NHWC buffers in the
compute dtype (ops.py).  ``encoder`` / ``decoder`` / ``model`` containers reproduce the
reference's slot indices, so checkpoints of either code base load into the other.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import _hip as H
from .. import ops
from .blocks import ModulatedResnetBlock, ResnetBlock, _Slot
from .layers import Conv2dWeightModulate, DownSample, EqualisedConv2d, EqualisedLinear, UpSample


class MappingNetwork(nn.Module):
    """z -> w MLP plus the style sampling helpers (reference builder.py:16-132).

    B x w_dim work: plain torch.  The CPU-RNG draw order of the reference is part of the
    contract (mix decision, crossover, z1, z2) and is kept exactly.
    """

    def __init__(self, features: int, n_layers: int, style_mixing_prob: float):
        super().__init__()
        self.d_latent = features
        self.style_mixing_prob = style_mixing_prob
        stack = []
        for i in range(n_layers):
            stack.append(EqualisedLinear(features, features))
            last = i == n_layers - 1
            # the final ReLU makes theta = 0 map to the zero style (reference builder.py:35-36)
            stack.append(nn.ReLU(inplace=True) if last else nn.LeakyReLU(0.2, inplace=True))
        self.net = nn.Sequential(*stack)
        self.register_buffer("shoeprint_style_vector", torch.zeros(1, 1, features), persistent=False)
        # True: every draw of the style path comes from the DEVICE generator and the style mixing is a mask, not a Python
        # branch -- the same distributions with no host dependence (a graph-capturable step: core/graphed.py).  Default:
        # the reference's CPU draws in the reference's order.
        self.device_draws = False

    def forward(self, z: torch.Tensor):
        return self.net(F.normalize(z.float(), dim=1))

    def _sample(self, batch_size, device):
        return self.forward(ops.host_to_device(torch.randn(batch_size, self.d_latent), device))

    def _get_style_vector(self, batch_size, n_gen_blocks, device, *, mix_styles=True):
        if self.device_draws:
            # both latents always go through the MLP; blocks [0, cut) take the first, the rest the second when the step
            # mixes (probability style_mixing_prob), all blocks the first when it does not
            z = torch.randn(2 * batch_size, self.d_latent, device=device)
            first, second = self.forward(z).split(batch_size, 0)
            mix = torch.rand((), device=device) < (self.style_mixing_prob if mix_styles else -1.0)
            cut = torch.randint(0, n_gen_blocks, (), device=device)
            use_first = (torch.arange(n_gen_blocks, device=device) < cut) | ~mix
            return torch.where(use_first.view(-1, 1, 1), first.unsqueeze(0), second.unsqueeze(0))
        if mix_styles and torch.rand(()).lt(self.style_mixing_prob):
            cut = int(torch.randint(0, n_gen_blocks, ()))
            # the two latents of a mixed style through the MLP as ONE 2B-row pass (same draws, in the reference's
            # order; every row of the MLP is independent): half the tiny launches of this path
            z = torch.cat((torch.randn(batch_size, self.d_latent), torch.randn(batch_size, self.d_latent)), 0)
            first, second = self.forward(ops.host_to_device(z, device)).split(batch_size, 0)
            return torch.cat((first.expand(cut, -1, -1), second.expand(n_gen_blocks - cut, -1, -1)), 0)
        return self._sample(batch_size, device).expand(n_gen_blocks, -1, -1)

    def get_single_w(self, batch_size, n_gen_blocks, device, domain_variable, *, mix_styles=True):
        base = self.shoeprint_style_vector
        if domain_variable == 0:  # no RNG draw on this path (reference builder.py:87-90)
            return base.expand(n_gen_blocks, batch_size, self.d_latent)
        style = self._get_style_vector(batch_size, n_gen_blocks, device, mix_styles=mix_styles)
        if isinstance(domain_variable, torch.Tensor):
            theta = domain_variable.view(1, -1, 1)
        else:
            theta = torch.full((1, 1, 1), float(domain_variable), device=device)
        return torch.lerp(base, style, theta)

    def get_two_w(self, batch_size, n_gen_blocks, device, domain_variables, *, mix_styles=True):
        style = self._get_style_vector(batch_size, n_gen_blocks, device, mix_styles=mix_styles)
        base = self.shoeprint_style_vector
        return tuple(torch.lerp(base, style, d.view(1, -1, 1)) for d in domain_variables)


class Generator(nn.Module):
    """Encoder (InstanceNorm ResNet) + weight-modulated decoder (reference builder.py:138-253)."""

    def __init__(self, input_nc: int, w_dim: int, image_size, min_latent_resolution: int,
                 n_resnet_blocks: int, start_filters: int = 64):
        super().__init__()
        self.input_nc = input_nc
        f = start_filters
        n_down = math.ceil(math.log2(min(image_size) / min_latent_resolution))
        enc = [_Slot("ReflectionPad2d(3) -> conv loader"), EqualisedConv2d(input_nc, f, kernel_size=7),
               _Slot("InstanceNorm2d"), _Slot("ReLU")]
        for _ in range(n_down):
            enc += [EqualisedConv2d(f, 2 * f, kernel_size=3, padding=1), _Slot("InstanceNorm2d"),
                    _Slot("ReLU"), DownSample()]
            f *= 2
        enc += [ResnetBlock(f) for _ in range(n_resnet_blocks // 2)]
        self.encoder = nn.Sequential(*enc)
        self.latent_nc = f

        dec = [ModulatedResnetBlock(f, w_dim=w_dim) for _ in range(math.ceil(n_resnet_blocks / 2))]
        for _ in range(n_down):
            dec += [UpSample(), Conv2dWeightModulate(f, f // 2, kernel_size=3, padding=1, w_dim=w_dim),
                    _Slot("ReLU -> conv epilogue")]
            f //= 2
        dec += [_Slot("ReflectionPad2d(3) -> conv loader"), EqualisedConv2d(f, input_nc, kernel_size=7),
                _Slot("Tanh -> conv epilogue")]
        self.decoder = nn.ModuleList(dec)
        self.n_style_blocks = sum(
            [isinstance(m, (ModulatedResnetBlock, Conv2dWeightModulate)) for m in self.decoder])

    # ---- fused plans on internal buffers -------------------------------------------------
    def _encode(self, t):
        mods = list(self.encoder)
        t = mods[1].run_norm_act(t, reflect=3, act=H.ACT_RELU)
        k = 4
        while k < len(mods):
            m = mods[k]
            if isinstance(m, EqualisedConv2d):  # conv, norm, relu, DownSample: slots k .. k + 3
                t = m.run_norm_act_down(t, mods[k + 3], act=H.ACT_RELU)
                k += 4
            else:
                t = m.run(t)  # ResnetBlock
                k += 1
        return t

    def _decode(self, t, w, collect: bool, internal: bool = False, tap=None):
        """``internal``: hand back NHWC buffers instead of public views (core/training.py runs several
        decodes as one batch and splits the result itself).  ``tap`` (with ``collect`` and ``internal``): per-sample
        weights of the path-loss pair term; the collected entries are then (term, channels, map shape) with the term
        taken as the map passes (ops.halves_sq_tap) instead of the maps themselves."""
        feats, i = [], 0
        # one UnbindBackward (a single stack) instead of a zeros + copy + accumulate trio per style block
        w = w.unbind(0)
        for m in self.decoder:
            if isinstance(m, ModulatedResnetBlock):
                t = m.run(t, w[i])
            elif isinstance(m, Conv2dWeightModulate):
                # extract() returns its last map BEFORE the ReLU; the earlier ones come back
                # post-ReLU because the reference's in-place ReLU aliases them (builder.py:196,241)
                last = collect and i + 1 == self.n_style_blocks
                t = m.run(t, w[i], act=H.ACT_NONE if last else H.ACT_RELU)
            elif isinstance(m, UpSample):
                t = m.run(t)
                continue
            elif isinstance(m, EqualisedConv2d):
                if collect:
                    break
                y = m.run(t, reflect=3, act=H.ACT_TANH)
                return y if internal else ops.to_public(y, self.input_nc)
            else:
                continue
            i += 1
            if collect:
                c = m.out_features if isinstance(m, Conv2dWeightModulate) else m.dim
                if tap is not None and internal:
                    t, term = ops.halves_sq_tap(t, tap)
                    feats.append((term, c, tuple(t.shape)))
                else:
                    feats.append((t, c) if internal else ops.to_public(t, c))
                if i == self.n_style_blocks:
                    return feats
        raise ValueError("No return layers specified.")

    def prepare_decoder_weights(self):
        """Kernel-side forms of every decoder filter, on the CURRENT stream (before a step forks a second
        stream that uses them too; a no-op when they are cached for the current parameter version)."""
        for m in self.decoder.modules():
            if isinstance(m, (EqualisedConv2d, Conv2dWeightModulate)):
                m._prepared().get()

    # ---- reference API -----------------------------------------------------------------------
    def encode(self, x: torch.Tensor):
        """Encode x to latent space z."""
        return ops.to_public(self._encode(ops.to_internal(x)), self.latent_nc)

    def decode(self, z: torch.Tensor, w: torch.Tensor):
        """Decode from latent space z to image, using style vector w."""
        return self._decode(ops.to_internal(z), w, collect=False)

    def extract(self, z: torch.Tensor, w: torch.Tensor):
        """Return the feature map behind every style block."""
        return self._decode(ops.to_internal(z), w, collect=True)

    def forward(self, x: torch.Tensor, w: torch.Tensor):
        return self._decode(self._encode(ops.to_internal(x)), w, collect=False)


def _patch_trunk(input_nc: int):
    return [
        EqualisedConv2d(input_nc, 64, kernel_size=4, padding=1), _Slot("LeakyReLU(0.2) -> conv epilogue"),
        DownSample(),
        EqualisedConv2d(64, 128, kernel_size=4, padding=1), _Slot("InstanceNorm2d"), _Slot("LeakyReLU(0.2)"),
        DownSample(),
        EqualisedConv2d(128, 256, kernel_size=4, padding=1), _Slot("InstanceNorm2d"), _Slot("LeakyReLU(0.2)"),
        DownSample(),
        EqualisedConv2d(256, 512, kernel_size=4, padding=1), _Slot("InstanceNorm2d"), _Slot("LeakyReLU(0.2)"),
    ]


def _run_trunk(mods, t):
    t = mods[2].run(mods[0].run(t, act=H.ACT_LRELU))
    t = mods[3].run_norm_act_down(t, mods[6], act=H.ACT_LRELU)
    t = mods[7].run_norm_act_down(t, mods[10], act=H.ACT_LRELU)
    return mods[11].run_norm_act(t, act=H.ACT_LRELU)


class Discriminator(nn.Module):
    """PatchGAN with stride-1 4x4 convs and blur-downsampling (reference builder.py:259-287)."""

    def __init__(self, input_nc: int):
        super().__init__()
        self.model = nn.Sequential(*_patch_trunk(input_nc), EqualisedConv2d(512, 1, kernel_size=4, padding=1))

    def forward(self, x: torch.Tensor):
        t = _run_trunk(self.model, ops.to_internal(x))
        return ops.to_public(self.model[14].run(t), 1)


class StyleExtractor(nn.Module):
    """Discriminator trunk -> global average pool -> linear (reference builder.py:293-320)."""

    def __init__(self, input_nc: int = 1, w_dim: int = 8):
        super().__init__()
        self.model = nn.Sequential(*_patch_trunk(input_nc), _Slot("AdaptiveAvgPool2d(1)"), _Slot("Flatten"),
                                   EqualisedLinear(512, w_dim))

    def forward(self, x):
        t = _run_trunk(self.model, ops.to_internal(x))
        return self.model[16](t.mean(dim=(1, 2), dtype=torch.float32))
