"""Oracle networks: plain-torch fp32 restatement of the reference model (TEST ONLY).

Every function cites the reference lines it restates (paths relative to /root/reference).
The module tree reproduces the reference ``state_dict`` keys so
that one closed-form weight fill (oracle/detweights.py) drives the reference, the oracle
and the HIP path alike.  The arithmetic is written functionally (``f_*`` helpers) and the
``nn.Module`` classes only hold parameters and call the helpers.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

# --------------------------------------------------------------------------------------
# functional restatement of src/model/layers.py
# --------------------------------------------------------------------------------------


def he_constant(shape) -> float:
    """layers.py:19 -- equalised learning-rate constant 1/sqrt(fan_in)."""
    fan_in = 1
    for d in shape[1:]:
        fan_in *= int(d)
    return 1.0 / math.sqrt(fan_in)


def f_eq_conv2d(x, weight, bias, padding):
    """layers.py:82-100 -- conv with runtime-scaled weight, stride 1, dilation 1."""
    return F.conv2d(x, weight * he_constant(weight.shape), bias, stride=1, padding=padding)


def f_eq_linear(x, weight, bias):
    """layers.py:39-40."""
    return F.linear(x, weight * he_constant(weight.shape), bias)


def f_modconv(x, w_style, weight, style_weight, style_bias, padding, eps=1e-8):
    """layers.py:145-182 -- modulate per input channel, demodulate per (sample, out channel),
    run as one grouped conv with per-sample weights.  No bias / noise / activation."""
    n, _, h, w = x.shape
    cout, cin, k, _ = weight.shape
    s = f_eq_linear(w_style, style_weight, style_bias)  # (n, cin); to_style bias is 1
    per_sample = (weight * he_constant(weight.shape)).unsqueeze(0) * s.view(n, 1, cin, 1, 1)
    inv_sigma = torch.rsqrt(per_sample.square().sum(dim=(2, 3, 4), keepdim=True) + eps)
    per_sample = per_sample * inv_sigma
    y = F.conv2d(
        x.reshape(1, n * cin, h, w),
        per_sample.reshape(n * cout, cin, k, k),
        padding=padding,
        groups=n,
    )
    return y.reshape(n, cout, y.shape[-2], y.shape[-1])


_BLUR = torch.tensor([[1.0, 2.0, 1.0], [2.0, 4.0, 2.0], [1.0, 2.0, 1.0]]) / 16.0


def f_blur(x, kernel=None):
    """layers.py:207-214 -- replicate-pad 1 then depthwise 3x3 binomial blur."""
    n, c, h, w = x.shape
    k = (_BLUR if kernel is None else kernel).to(x).view(1, 1, 3, 3)
    y = F.conv2d(F.pad(x.reshape(n * c, 1, h, w), (1, 1, 1, 1), mode="replicate"), k)
    return y.view(n, c, h, w)


def f_upsample(x, kernel=None):
    """layers.py:223-229 -- bilinear x2 (align_corners=False) then blur."""
    return f_blur(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), kernel)


def f_downsample(x, kernel=None):
    """layers.py:241-247 -- blur then bilinear resize to floor(H/2) x floor(W/2)."""
    x = f_blur(x, kernel)
    return F.interpolate(
        x, (x.shape[2] // 2, x.shape[3] // 2), mode="bilinear", align_corners=False
    )


def f_instance_norm(x):
    """nn.InstanceNorm2d defaults: eps 1e-5, biased variance, no affine, batch stats."""
    return F.instance_norm(x, eps=1e-5)


def f_reflect(x, p):
    return F.pad(x, (p, p, p, p), mode="reflect")


# --------------------------------------------------------------------------------------
# parameter holders with the reference's state_dict layout
# --------------------------------------------------------------------------------------


class _EqW(nn.Module):
    """Holds ``weight`` ~ N(0,1) (layers.py:21); key ``<path>.weight.weight``."""

    def __init__(self, shape):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(*shape))


class EqLinear(nn.Module):
    def __init__(self, fin, fout, bias=0.0):
        super().__init__()
        self.weight = _EqW((fout, fin))
        self.bias = nn.Parameter(torch.full((fout,), float(bias)))

    def forward(self, x):
        return f_eq_linear(x, self.weight.weight, self.bias)


class EqConv(nn.Module):
    def __init__(self, cin, cout, k, padding=0, use_bias=True):
        super().__init__()
        self.padding = padding
        self.weight = _EqW((cout, cin, k, k))
        if use_bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.bias = None

    def forward(self, x):
        return f_eq_conv2d(x, self.weight.weight, self.bias, self.padding)


class ModConv(nn.Module):
    def __init__(self, cin, cout, k, w_dim, padding):
        super().__init__()
        self.padding = padding
        self.weight = _EqW((cout, cin, k, k))
        self.to_style = EqLinear(w_dim, cin, bias=1.0)

    def forward(self, x, w):
        return f_modconv(
            x, w, self.weight.weight, self.to_style.weight.weight, self.to_style.bias, self.padding
        )


class _Smooth(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("kernel", _BLUR.clone().view(1, 1, 3, 3))


class Up(nn.Module):
    def __init__(self):
        super().__init__()
        self.smooth = _Smooth()

    def forward(self, x):
        return f_upsample(x, self.smooth.kernel)


class Down(nn.Module):
    def __init__(self):
        super().__init__()
        self.smooth = _Smooth()

    def forward(self, x):
        return f_downsample(x, self.smooth.kernel)


class _Fn(nn.Module):
    """Parameter-free step occupying one ``Sequential`` slot (keeps the indices aligned)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


# Test hook (tests/test_hip_parity.py, shared-mask gradient check): when set to a callable
# ``(kind, tensor, inplace) -> tensor`` it replaces every ReLU / LeakyReLU of the oracle networks, e.g.
# to replay the activation masks recorded from another run.  None = the plain functions.
ACT_OVERRIDE = None


def _relu(inplace=False):
    plain = torch.relu_ if inplace else torch.relu
    return _Fn(lambda t: plain(t) if ACT_OVERRIDE is None else ACT_OVERRIDE("relu", t, inplace))


def _lrelu():
    return _Fn(lambda t: F.leaky_relu(t, 0.2) if ACT_OVERRIDE is None else ACT_OVERRIDE("lrelu", t, False))


def _inorm():
    return _Fn(f_instance_norm)


def _rpad(p):
    return _Fn(lambda t: f_reflect(t, p))


class ResBlock(nn.Module):
    """blocks.py:9-33 -- x + IN(conv(rpad(ReLU(IN(conv(rpad(x))))))); convs have no bias."""

    def __init__(self, dim):
        super().__init__()
        self.conv_block = nn.Sequential(
            _rpad(1), EqConv(dim, dim, 3, 0, use_bias=False), _inorm(), _relu(),
            _rpad(1), EqConv(dim, dim, 3, 0, use_bias=False), _inorm(),
        )

    def forward(self, x):
        return x + self.conv_block(x)


class ModResBlock(nn.Module):
    """blocks.py:36-68 -- x + modconv(rpad(ReLU(modconv(rpad(x), w))), w); same w twice."""

    def __init__(self, dim, w_dim):
        super().__init__()
        self.conv_block = nn.ModuleList(
            [_rpad(1), ModConv(dim, dim, 3, w_dim, 0), _relu(), _rpad(1), ModConv(dim, dim, 3, w_dim, 0)]
        )

    def forward(self, x, w):
        y = x
        for m in self.conv_block:
            y = m(y, w) if isinstance(m, ModConv) else m(y)
        return x + y


# --------------------------------------------------------------------------------------
# networks: src/model/builder.py
# --------------------------------------------------------------------------------------


class MappingNetwork(nn.Module):
    """builder.py:16-132."""

    def __init__(self, features, n_layers, style_mixing_prob):
        super().__init__()
        self.d_latent = features
        self.style_mixing_prob = style_mixing_prob
        mods = []
        for i in range(n_layers):
            mods.append(EqLinear(features, features))
            mods.append(_relu() if i == n_layers - 1 else _lrelu())  # builder.py:35-36
        self.net = nn.Sequential(*mods)
        self.register_buffer(
            "shoeprint_style_vector", torch.zeros(1, 1, features), persistent=False
        )

    def forward(self, z):
        return self.net(F.normalize(z, dim=1))  # builder.py:46-49

    def _get_style_vector(self, batch_size, n_gen_blocks, device, *, mix_styles=True):
        # builder.py:106-132: CPU-RNG draw order is rand() -> randint -> randn -> randn
        if mix_styles and bool(torch.rand(()) < self.style_mixing_prob):
            cross = int(torch.randint(0, n_gen_blocks, ()))
            s1 = self.forward(torch.randn(batch_size, self.d_latent).to(device))
            s2 = self.forward(torch.randn(batch_size, self.d_latent).to(device))
            return torch.cat(
                (s1.unsqueeze(0).expand(cross, -1, -1),
                 s2.unsqueeze(0).expand(n_gen_blocks - cross, -1, -1)), dim=0)
        s = self.forward(torch.randn(batch_size, self.d_latent).to(device))
        return s.unsqueeze(0).expand(n_gen_blocks, -1, -1)

    def get_single_w(self, batch_size, n_gen_blocks, device, domain_variable, *, mix_styles=True):
        zero = self.shoeprint_style_vector
        if bool(domain_variable == 0):  # multi-element tensors raise, as in the reference
            return zero.expand(n_gen_blocks, batch_size, self.d_latent)  # builder.py:87-90
        s = self._get_style_vector(batch_size, n_gen_blocks, device, mix_styles=mix_styles)
        if isinstance(domain_variable, torch.Tensor):
            d = domain_variable.view(1, -1, 1)
        else:
            d = torch.tensor(float(domain_variable), device=device).view(1, 1, 1)
        return torch.lerp(zero, s, d)  # builder.py:104

    def get_two_w(self, batch_size, n_gen_blocks, device, domain_variables, *, mix_styles=True):
        d1, d2 = domain_variables
        s = self._get_style_vector(batch_size, n_gen_blocks, device, mix_styles=mix_styles)
        zero = self.shoeprint_style_vector
        return torch.lerp(zero, s, d1.view(1, -1, 1)), torch.lerp(zero, s, d2.view(1, -1, 1))


class Generator(nn.Module):
    """builder.py:138-253."""

    def __init__(self, input_nc, w_dim, image_size, min_latent_resolution, n_resnet_blocks,
                 start_filters=64):
        super().__init__()
        f = start_filters
        n_down = math.ceil(math.log2(min(image_size) / min_latent_resolution))
        enc = [_rpad(3), EqConv(input_nc, f, 7), _inorm(), _relu()]
        for _ in range(n_down):
            enc += [EqConv(f, 2 * f, 3, 1), _inorm(), _relu(), Down()]
            f *= 2
        enc += [ResBlock(f) for _ in range(n_resnet_blocks // 2)]
        self.encoder = nn.Sequential(*enc)
        dec = [ModResBlock(f, w_dim) for _ in range(math.ceil(n_resnet_blocks / 2))]
        for _ in range(n_down):
            # nn.ReLU(inplace=True) (builder.py:196): it aliases the map that extract()
            # appended one line earlier, so every non-final Conv2dWeightModulate feature
            # is returned post-ReLU; the final one is returned before the ReLU runs.
            dec += [Up(), ModConv(f, f // 2, 3, w_dim, 1), _relu(inplace=True)]
            f //= 2
        dec += [_rpad(3), EqConv(f, input_nc, 7), _Fn(torch.tanh)]
        self.decoder = nn.ModuleList(dec)
        self.n_style_blocks = sum(isinstance(m, (ModResBlock, ModConv)) for m in self.decoder)

    def encode(self, x):
        return self.encoder(x)

    def _walk(self, z, w, collect):
        i = 0
        feats = []
        for m in self.decoder:
            if isinstance(m, (ModResBlock, ModConv)):
                z = m(z, w[i])
                i += 1
                if collect:
                    feats.append(z)
                    if i == self.n_style_blocks:
                        return feats
            else:
                z = m(z)
        if collect:
            raise ValueError("No return layers specified.")  # builder.py:248-249
        return z

    def decode(self, z, w):
        return self._walk(z, w, collect=False)

    def extract(self, z, w):
        return self._walk(z, w, collect=True)

    def forward(self, x, w):
        return self.decode(self.encode(x), w)


def _patch_trunk(input_nc):
    """builder.py:268-282 / 299-313 -- the shared D / S trunk (indices 0..13)."""
    return [
        EqConv(input_nc, 64, 4, 1), _lrelu(), Down(),
        EqConv(64, 128, 4, 1), _inorm(), _lrelu(), Down(),
        EqConv(128, 256, 4, 1), _inorm(), _lrelu(), Down(),
        EqConv(256, 512, 4, 1), _inorm(), _lrelu(),
    ]


class Discriminator(nn.Module):
    """builder.py:259-287."""

    def __init__(self, input_nc):
        super().__init__()
        self.model = nn.Sequential(*_patch_trunk(input_nc), EqConv(512, 1, 4, 1))

    def forward(self, x):
        return self.model(x)


class StyleExtractor(nn.Module):
    """builder.py:293-320."""

    def __init__(self, input_nc=1, w_dim=8):
        super().__init__()
        self.model = nn.Sequential(
            *_patch_trunk(input_nc),
            _Fn(lambda t: t.mean(dim=(2, 3), keepdim=True)),  # AdaptiveAvgPool2d(1)
            _Fn(lambda t: t.flatten(1)),
            EqLinear(512, w_dim),
        )

    def forward(self, x):
        return self.model(x)
