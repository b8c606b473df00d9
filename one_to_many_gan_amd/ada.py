"""Adaptive discriminator augmentation on the GPU.

Interface of ``ada.AdaptiveDiscriminatorAugmentation`` as the reference uses it
(train.py:175-188 constructor switches, train.py:206 ``set_p``, training.py:100,104,200 call on
(B, C, H, W) discriminator inputs; differentiable with respect to the images).  The dependency
(pytorch-ada @ 99754cb4) is not vendored, so the transforms follow the published StyleGAN2-ADA
pipe (Karras et al. 2020, Appendix B) -- **parity unpinned**, see oracle/ada.py.

MI355X design
* The per-image 3x3 geometry and 4x4 colour matrices are sampled and composed on the HOST in
  float64 (a few hundred flops per image; the CPU has ~30 ms of slack per step) and reach the GPU
  as one small tensor each -- not ~80 tiny device kernels.
* Geometry = [reflect pad + 2x sym6 upsample] -> bilinear affine resampling -> [2x sym6
  downsample + crop].  Both bracketed stages are banded separable linear operators, so they run
  through o2m_resample2d with operators composed on the host (reflection maps a contiguous window
  onto a contiguous window, so padding folds into the upsampling operator); the 12-tap stages run
  as a vertical and a horizontal 1-D pass.  The resampling in the middle is o2m_ada_grid_sample.
* Backward = the adjoints in reverse: transposed banded operators, o2m_ada_grid_sample_bwd (a
  gather: each source pixel enumerates the few outputs that sample it), o2m_reflect_fold.
* Colour = one per-pixel 3x4 affine (o2m_ada_colour); its adjoint is the transposed matrix.
"""

from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import _hip as H
from . import ops
from . import resample as R

SYM6 = np.array([0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
                 0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
                 -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148])
HZ_PAD = len(SYM6) // 4  # 3

# (name, distribution, values per image) in the order the pipe consumes them
_DRAWS = (("xflip_i", "u", 1), ("xflip_g", "u", 1), ("rot90_i", "u", 1), ("rot90_g", "u", 1),
          ("xint_t", "u", 2), ("xint_g", "u", 1), ("scale_s", "n", 1), ("scale_g", "u", 1),
          ("rot1_t", "u", 1), ("rot1_g", "u", 1), ("aniso_s", "n", 1), ("aniso_g", "u", 1),
          ("rot2_t", "u", 1), ("rot2_g", "u", 1), ("xfrac_t", "n", 2), ("xfrac_g", "u", 1),
          ("bright_b", "n", 1), ("bright_g", "u", 1), ("contr_c", "n", 1), ("contr_g", "u", 1),
          ("luma_i", "u", 1), ("luma_g", "u", 1), ("hue_t", "u", 1), ("hue_g", "u", 1),
          ("sat_s", "n", 1), ("sat_g", "u", 1))


# ----------------------------------------------------------------------------- host algebra


def _eye(b, n):
    return np.tile(np.eye(n), (b, 1, 1))


def _trans(b, tx, ty):
    m = _eye(b, 3)
    m[:, 0, 2], m[:, 1, 2] = tx, ty
    return m


def _scale(b, sx, sy):
    m = _eye(b, 3)
    m[:, 0, 0], m[:, 1, 1] = sx, sy
    return m


def _rot(b, theta):
    m = _eye(b, 3)
    c, s = np.cos(theta), np.sin(theta)
    m[:, 0, 0], m[:, 0, 1], m[:, 1, 0], m[:, 1, 1] = c, -s, s, c
    return m


def _reflect(i, n):
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def _pad_up_coo(n, m0, m1):
    """Reflection padding (m0, m1) + zero insertion + 12-tap convolution (gain 2) as COO triplets
    of the [2L x n] operator, L = n + m0 + m1.  Output o meets the filter taps of its own parity:
    k = (o & 1) + 2j, padded sample i = (o + k - 6) / 2."""
    L = n + m0 + m1
    f = SYM6 / SYM6.sum()
    ff = (2.0 * f)[::-1]
    o = np.repeat(np.arange(2 * L), 6)
    k = (o & 1) + 2 * np.tile(np.arange(6), 2 * L)
    i = (o + k - 6) // 2
    ok = (i >= 0) & (i < L)
    return o[ok], _reflect(i[ok] - m0, n), ff[k[ok]], 2 * L, n


def _up_t_coo(L):
    """Adjoint of the 2x upsampling on the padded grid: [L x 2L], row i gathers outputs
    o = 2i + 6 - k with weight ff[k]."""
    f = SYM6 / SYM6.sum()
    ff = (2.0 * f)[::-1]
    i = np.repeat(np.arange(L), 12)
    k = np.tile(np.arange(12), L)
    o = 2 * i + 6 - k
    ok = (o >= 0) & (o < 2 * L)
    return i[ok], o[ok], ff[k[ok]], L, 2 * L


def _dense(op):
    rows, cols, vals, n_rows, n_cols = op
    a = np.zeros((n_rows, n_cols))
    np.add.at(a, (rows, cols), vals)
    return a


def _pad_up_operator(n, m0, m1):
    """Dense [2L x n] form of _pad_up_coo (tests, documentation)."""
    return _dense(_pad_up_coo(n, m0, m1))


def _up_operator(L):
    return _dense(_pad_up_coo(L, 0, 0))


def _down_operator(n):
    """[n x 2n+12]: crop 1, 12-tap correlation, every second sample."""
    f = SYM6 / SYM6.sum()
    a = np.zeros((n, 2 * n + 4 * HZ_PAD))
    for o in range(n):
        a[o, 1 + 2 * o: 1 + 2 * o + 12] = f
    return a


class _Plan:
    """Everything one geometric call needs on the device."""


def _pass(x, start, weights, t, span, axis):
    """One 1-D banded pass along ``axis`` (1 = vertical, 2 = horizontal) of an NHWC buffer; span 0
    selects the per-output kernel (arbitrary starts, 6..8 taps)."""
    B, Hh, Ww, Cn = x.shape
    n_out = weights.shape[0]
    # ``span`` is an upper bound on the step between tap starts; the instantiated 1-D kernels are
    # (12 taps, span 2) and (6 taps, span 1 or 2)
    if span > 0:
        span = max(span, 2 if t == 12 else 1)
    if axis == 1:
        ix, iw = R.identity_taps(Ww, x.device)
        y = torch.empty((B, n_out, Ww, Cn), dtype=x.dtype, device=x.device)
        H.resample2d(x, y, start, weights, ix, iw, t, 1, span, 1 if span else 0)
    else:
        iy, iw = R.identity_taps(Hh, x.device)
        y = torch.empty((B, Hh, n_out, Cn), dtype=x.dtype, device=x.device)
        H.resample2d(x, y, iy, iw, start, weights, 1, t, 1 if span else 0, span)
    return y


_down_cache: dict = {}


def _down_taps(n, device):
    key = (n, str(device))
    hit = _down_cache.get(key)
    if hit is None:
        a = _down_operator(n)
        hit = (R.taps_1d(a, device), R.taps_1d(a.T, device))
        _down_cache[key] = hit
    return hit


class _GeometryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, plan):
        # pad + upsample as a vertical and a horizontal 1-D pass (6..8 + 6..8 loads per output instead
        # of T^2; the starts are not monotonic across the mirror, so the per-output kernel)
        sy, wy, sx, wx, T, _, _ = plan.padup
        x1 = _pass(_pass(t, sy, wy, int(T), 0, 1), sx, wx, int(T), 0, 2)  # [B][2Ly][2Lx][Cp]
        B = t.shape[0]
        x2 = torch.empty((B, plan.ho, plan.wo, t.shape[3]), dtype=t.dtype, device=t.device)
        H.ada_grid_sample(x1, plan.theta, x2)
        (dy, _), (dx, _) = plan.down_y, plan.down_x
        y = _pass(_pass(x2, *dy, 1), *dx, 2)
        ctx.plan, ctx.src_shape = plan, x1.shape
        return y

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = g.contiguous()
        (_, dyt), (_, dxt) = plan.down_y, plan.down_x
        g2 = _pass(_pass(g, *dxt, 2), *dyt, 1)  # [B][Ho][Wo][Cp]
        g1 = torch.empty(ctx.src_shape, dtype=g.dtype, device=g.device)
        H.ada_grid_sample_bwd(g2, plan.theta, g1)
        gp = _pass(_pass(g1, *plan.up_yt, 1), *plan.up_xt, 2)  # padded grid [B][Ly][Lx][Cp]
        gx = torch.empty((g.shape[0], plan.h, plan.w, g.shape[3]), dtype=g.dtype, device=g.device)
        H.reflect_fold(gp, gx, plan.my0, plan.mx0)
        return gx, None


class _ColourFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, m, m_t, channels):
        y = torch.empty_like(t)
        H.ada_colour(t, m, y, channels)
        ctx.channels = channels
        ctx.save_for_backward(m_t)
        return y

    @staticmethod
    def backward(ctx, g):
        (m_t,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(g)
        H.ada_colour(g, m_t, gx, ctx.channels)
        return gx, None, None, None


class AdaptiveDiscriminatorAugmentation(nn.Module):
    """Drop-in for ``ada.AdaptiveDiscriminatorAugmentation``: the constructor switches are the
    strengths of the transform groups (probability = strength * p), ``set_p`` sets p."""

    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0,
                 scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0,
                 lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5, hue_max=1,
                 saturation_std=1, generator: torch.Generator | None = None):
        super().__init__()
        self.xflip, self.rotate90, self.xint, self.xint_max = float(xflip), float(rotate90), float(xint), xint_max
        self.scale, self.rotate, self.aniso, self.xfrac = float(scale), float(rotate), float(aniso), float(xfrac)
        self.scale_std, self.rotate_max, self.aniso_std, self.xfrac_std = scale_std, rotate_max, aniso_std, xfrac_std
        self.brightness, self.contrast, self.lumaflip = float(brightness), float(contrast), float(lumaflip)
        self.hue, self.saturation = float(hue), float(saturation)
        self.brightness_std, self.contrast_std, self.hue_max = brightness_std, contrast_std, hue_max
        self.saturation_std = saturation_std
        self.generator = generator  # CPU generator of the random draws (None: torch's default)
        self.p = 0.0

    def set_p(self, p: float):
        self.p = float(p)

    # ------------------------------------------------------------------ sampling (host)
    def draw(self, batch: int):
        out = {}
        for name, kind, n in _DRAWS:
            fn = torch.rand if kind == "u" else torch.randn
            out[name] = fn((batch, n), generator=self.generator, dtype=torch.float64).numpy()
        return out

    def geometry_matrix(self, d, width, height):
        """G_inv [B,3,3] float64: output pixel coordinates (centre origin) -> input coordinates."""
        b, p = d["xflip_i"].shape[0], self.p
        g = _eye(b, 3)
        if self.xflip > 0:
            i = np.where(d["xflip_g"][:, 0] < self.xflip * p, np.floor(d["xflip_i"][:, 0] * 2), 0.0)
            g = g @ _scale(b, 1.0 / (1.0 - 2.0 * i), 1.0)
        if self.rotate90 > 0:
            i = np.where(d["rot90_g"][:, 0] < self.rotate90 * p, np.floor(d["rot90_i"][:, 0] * 4), 0.0)
            g = g @ _rot(b, math.pi / 2 * i)
        if self.xint > 0:
            t = np.where(d["xint_g"] < self.xint * p, (d["xint_t"] * 2 - 1) * self.xint_max, 0.0)
            g = g @ _trans(b, -np.round(t[:, 0] * width), -np.round(t[:, 1] * height))
        if self.scale > 0:
            s = np.where(d["scale_g"][:, 0] < self.scale * p, np.exp2(d["scale_s"][:, 0] * self.scale_std), 1.0)
            g = g @ _scale(b, 1.0 / s, 1.0 / s)
        p_rot = 1.0 - math.sqrt(min(max(1.0 - self.rotate * p, 0.0), 1.0))
        if self.rotate > 0:
            th = np.where(d["rot1_g"][:, 0] < p_rot, (d["rot1_t"][:, 0] * 2 - 1) * math.pi * self.rotate_max, 0.0)
            g = g @ _rot(b, th)
        if self.aniso > 0:
            s = np.where(d["aniso_g"][:, 0] < self.aniso * p, np.exp2(d["aniso_s"][:, 0] * self.aniso_std), 1.0)
            g = g @ _scale(b, 1.0 / s, s)
        if self.rotate > 0:
            th = np.where(d["rot2_g"][:, 0] < p_rot, (d["rot2_t"][:, 0] * 2 - 1) * math.pi * self.rotate_max, 0.0)
            g = g @ _rot(b, th)
        if self.xfrac > 0:
            t = np.where(d["xfrac_g"] < self.xfrac * p, d["xfrac_t"] * self.xfrac_std, 0.0)
            g = g @ _trans(b, -t[:, 0] * width, -t[:, 1] * height)
        return g

    def colour_matrix(self, d, channels):
        b, p = d["bright_b"].shape[0], self.p
        c = _eye(b, 4)
        v = np.array([1.0, 1.0, 1.0, 0.0]) / math.sqrt(3)
        vv = np.outer(v, v)
        if self.brightness > 0:
            br = np.where(d["bright_g"][:, 0] < self.brightness * p, d["bright_b"][:, 0] * self.brightness_std, 0.0)
            m = _eye(b, 4)
            m[:, 0, 3] = m[:, 1, 3] = m[:, 2, 3] = br
            c = m @ c
        if self.contrast > 0:
            ct = np.where(d["contr_g"][:, 0] < self.contrast * p, np.exp2(d["contr_c"][:, 0] * self.contrast_std), 1.0)
            m = _eye(b, 4)
            m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = ct
            c = m @ c
        if self.lumaflip > 0:
            i = np.where(d["luma_g"][:, 0] < self.lumaflip * p, np.floor(d["luma_i"][:, 0] * 2), 0.0)
            c = (np.eye(4)[None] - 2.0 * vv[None] * i[:, None, None]) @ c
        if self.hue > 0 and channels > 1:
            th = np.where(d["hue_g"][:, 0] < self.hue * p, (d["hue_t"][:, 0] * 2 - 1) * math.pi * self.hue_max, 0.0)
            k = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])  # cross-product matrix
            r3 = (np.cos(th)[:, None, None] * np.eye(3)[None] + np.sin(th)[:, None, None] * k[None]
                  + (1 - np.cos(th))[:, None, None] * np.outer(v[:3], v[:3])[None])  # Rodrigues
            m = _eye(b, 4)
            m[:, :3, :3] = r3
            c = m @ c
        if self.saturation > 0 and channels > 1:
            s = np.where(d["sat_g"][:, 0] < self.saturation * p, np.exp2(d["sat_s"][:, 0] * self.saturation_std), 1.0)
            c = (vv[None] + (np.eye(4) - vv)[None] * s[:, None, None]) @ c
        return c

    @staticmethod
    def margins(g_inv, width, height):
        cx, cy = (width - 1) / 2, (height - 1) / 2
        corners = np.array([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]]).T  # [xyz][idx]
        cp = g_inv @ corners  # [B][xyz][idx]
        lo = np.array([cp[:, 0, :].min(), cp[:, 1, :].min()])
        hi = np.array([cp[:, 0, :].max(), cp[:, 1, :].max()])
        m = np.concatenate([-lo, hi]) + np.array([HZ_PAD * 2 - cx, HZ_PAD * 2 - cy] * 2)
        m = np.minimum(np.maximum(m, 0.0), np.array([width - 1, height - 1] * 2))
        return [int(v) for v in np.ceil(m)]  # mx0, my0, mx1, my1

    def plan_geometry(self, g_inv, channels, height, width, device):
        b = g_inv.shape[0]
        mx0, my0, mx1, my1 = self.margins(g_inv, width, height)
        ly, lx = height + my0 + my1, width + mx0 + mx1
        hs, ws = 2 * ly, 2 * lx
        ho, wo = 2 * (height + 2 * HZ_PAD), 2 * (width + 2 * HZ_PAD)
        g = _trans(b, (mx0 - mx1) / 2, (my0 - my1) / 2) @ g_inv
        g = _scale(b, 2, 2) @ g @ _scale(b, 0.5, 0.5)
        g = _trans(b, -0.5, -0.5) @ g @ _trans(b, 0.5, 0.5)
        g = _scale(b, 2 / ws, 2 / hs) @ g @ _scale(b, wo / 2, ho / 2)
        plan = _Plan()
        plan.theta = torch.from_numpy(np.ascontiguousarray(g[:, :2, :].reshape(b, 6)).astype(np.float32)).to(device)
        plan.padup = R.taps_from_coo(_pad_up_coo(height, my0, my1), _pad_up_coo(width, mx0, mx1), device, min_width=6)
        if not 6 <= int(plan.padup[4]) <= 8:
            raise RuntimeError("padding + upsampling operator outside the instantiated 6..8 taps")
        plan.up_yt = R.taps_1d_coo(_up_t_coo(ly), device)
        plan.up_xt = R.taps_1d_coo(_up_t_coo(lx), device)
        plan.down_y, plan.down_x = _down_taps(height, device), _down_taps(width, device)
        plan.h, plan.w, plan.ho, plan.wo = height, width, ho, wo
        plan.mx0, plan.my0, plan.channels = mx0, my0, channels
        return plan

    @staticmethod
    def colour_operands(cm, channels, device):
        """(m, m_transposed) fp32 [B][3][4] for o2m_ada_colour and its adjoint."""
        b = cm.shape[0]
        m = np.zeros((b, 3, 4))
        mt = np.zeros((b, 3, 4))
        if channels == 3:
            m[:] = cm[:, :3, :]
            mt[:, :, :3] = np.transpose(cm[:, :3, :3], (0, 2, 1))
        else:  # grey: the mean luma response of the three rows
            row = cm[:, :3, :].mean(axis=1)  # [B][4]
            m[:, 0, 0], m[:, 0, 3] = row[:, :3].sum(axis=1), row[:, 3]
            mt[:, 0, 0] = m[:, 0, 0]
        to = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)  # noqa: E731
        return to(m), to(mt)

    # ------------------------------------------------------------------------- forward
    def forward(self, images: torch.Tensor, draws=None):
        if self.p == 0.0:
            return images
        b, c, h, w = images.shape
        if c not in (1, 3):
            raise ValueError("the colour transforms are defined for 1 or 3 channels")
        d = draws if draws is not None else self.draw(b)
        g_inv = self.geometry_matrix(d, w, h)
        cm = self.colour_matrix(d, c)
        t = ops.to_internal(images)
        if not np.array_equal(g_inv, _eye(b, 3)):
            t = _GeometryFn.apply(t, self.plan_geometry(g_inv, c, h, w, t.device))
        if not np.array_equal(cm, _eye(b, 4)):
            m, m_t = self.colour_operands(cm, c, t.device)
            t = _ColourFn.apply(t, m, m_t, c)
        return ops.to_public(t, c)
