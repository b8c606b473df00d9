"""CPU restatement of the adaptive-discriminator-augmentation pipe -- TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; the product is one_to_many_gan_amd/ada.py + csrc/ada.hip).

The reference calls ``ada.AdaptiveDiscriminatorAugmentation(xflip=1, rotate90=1, xint=1, scale=1,
rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)``
(train.py:175-188) on discriminator inputs (training.py:100,104,200) and ``set_p`` once per step
(train.py:206).  ``ada`` is the un-vendored dependency pytorch-ada @ 99754cb4 (uv.lock:762-764),
a packaging of the StyleGAN2-ADA augmentation pipe; its source is absent from /root/reference and
the reference holds no tests or fixtures for it.  This file restates the PUBLISHED algorithm
(Karras et al. 2020, "Training Generative Adversarial Networks with Limited Data", Appendix B:
the geometric and colour transform groups, sym6 low-pass up/down-sampling around one bilinear
resampling, reflection padding by the transformed corner margins) for exactly those switches.

Parity status: **unpinned** -- there is nothing of the dependency in this image to check against;
DESIGN.md says so.  The random draws are passed in explicitly (``draws``) so that the product
and this restatement can be driven by the same numbers.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]

# order and shape of the random numbers one call consumes (B = batch size); "u" uniform [0,1), "n" normal
DRAWS = [("xflip_i", "u", 1), ("xflip_g", "u", 1), ("rot90_i", "u", 1), ("rot90_g", "u", 1),
         ("xint_t", "u", 2), ("xint_g", "u", 1), ("scale_s", "n", 1), ("scale_g", "u", 1),
         ("rot1_t", "u", 1), ("rot1_g", "u", 1), ("aniso_s", "n", 1), ("aniso_g", "u", 1),
         ("rot2_t", "u", 1), ("rot2_g", "u", 1), ("xfrac_t", "n", 2), ("xfrac_g", "u", 1),
         ("bright_b", "n", 1), ("bright_g", "u", 1), ("contr_c", "n", 1), ("contr_g", "u", 1),
         ("luma_i", "u", 1), ("luma_g", "u", 1), ("hue_t", "u", 1), ("hue_g", "u", 1),
         ("sat_s", "n", 1), ("sat_g", "u", 1)]


def make_draws(batch: int, generator: torch.Generator):
    out = {}
    for name, kind, n in DRAWS:
        fn = torch.rand if kind == "u" else torch.randn
        out[name] = fn((batch, n), generator=generator, dtype=torch.float64)
    return out


def _eye(b, n):
    return torch.eye(n, dtype=torch.float64).repeat(b, 1, 1)


def _m3(b, rows):
    m = torch.zeros((b, 3, 3), dtype=torch.float64)
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            m[:, i, j] = v
    return m


def translate2d(b, tx, ty):
    return _m3(b, [[1, 0, tx], [0, 1, ty], [0, 0, 1]])


def scale2d(b, sx, sy):
    return _m3(b, [[sx, 0, 0], [0, sy, 0], [0, 0, 1]])


def rotate2d(b, theta):
    c, s = torch.cos(theta), torch.sin(theta)
    return _m3(b, [[c, -s, 0], [s, c, 0], [0, 0, 1]])


def geometry_matrix(draws, p, width, height, xint_max=0.125, scale_std=0.2, rotate_max=1.0, aniso_std=0.2,
                    xfrac_std=0.125):
    """G_inv [B,3,3]: maps OUTPUT pixel coordinates (centre origin) to input coordinates."""
    b = draws["xflip_i"].shape[0]
    g = _eye(b, 3)
    one, zero = torch.ones(b, dtype=torch.float64), torch.zeros(b, dtype=torch.float64)

    def gate(name, prob):
        return draws[name][:, 0] < prob

    i = torch.where(gate("xflip_g", p), torch.floor(draws["xflip_i"][:, 0] * 2), zero)
    g = g @ scale2d(b, 1 / (1 - 2 * i), one)
    i = torch.where(gate("rot90_g", p), torch.floor(draws["rot90_i"][:, 0] * 4), zero)
    g = g @ rotate2d(b, math.pi / 2 * i)  # rotate2d_inv(-pi/2 * i)
    t = (draws["xint_t"] * 2 - 1) * xint_max
    t = torch.where(gate("xint_g", p)[:, None], t, torch.zeros_like(t))
    g = g @ translate2d(b, -torch.round(t[:, 0] * width), -torch.round(t[:, 1] * height))
    s = torch.where(gate("scale_g", p), torch.exp2(draws["scale_s"][:, 0] * scale_std), one)
    g = g @ scale2d(b, 1 / s, 1 / s)
    p_rot = 1 - math.sqrt(min(max(1 - p, 0.0), 1.0))
    th = torch.where(gate("rot1_g", p_rot), (draws["rot1_t"][:, 0] * 2 - 1) * math.pi * rotate_max, zero)
    g = g @ rotate2d(b, th)  # rotate2d_inv(-theta)
    s = torch.where(gate("aniso_g", p), torch.exp2(draws["aniso_s"][:, 0] * aniso_std), one)
    g = g @ scale2d(b, 1 / s, s)
    th = torch.where(gate("rot2_g", p_rot), (draws["rot2_t"][:, 0] * 2 - 1) * math.pi * rotate_max, zero)
    g = g @ rotate2d(b, th)
    t = torch.where(gate("xfrac_g", p)[:, None], draws["xfrac_t"] * xfrac_std, torch.zeros_like(draws["xfrac_t"]))
    g = g @ translate2d(b, -t[:, 0] * width, -t[:, 1] * height)
    return g


def colour_matrix(draws, p, channels, brightness_std=0.2, contrast_std=0.5, hue_max=1.0, saturation_std=1.0):
    """C [B,4,4] acting on (r, g, b, 1)."""
    b = draws["bright_b"].shape[0]
    c = _eye(b, 4)
    one, zero = torch.ones(b, dtype=torch.float64), torch.zeros(b, dtype=torch.float64)
    v = torch.tensor([1.0, 1.0, 1.0, 0.0], dtype=torch.float64) / math.sqrt(3)
    vv = torch.outer(v, v)
    eye4 = torch.eye(4, dtype=torch.float64)

    def gate(name):
        return draws[name][:, 0] < p

    br = torch.where(gate("bright_g"), draws["bright_b"][:, 0] * brightness_std, zero)
    m = _eye(b, 4)
    m[:, 0, 3] = br; m[:, 1, 3] = br; m[:, 2, 3] = br
    c = m @ c
    ct = torch.where(gate("contr_g"), torch.exp2(draws["contr_c"][:, 0] * contrast_std), one)
    m = _eye(b, 4)
    m[:, 0, 0] = ct; m[:, 1, 1] = ct; m[:, 2, 2] = ct
    c = m @ c
    i = torch.where(gate("luma_g"), torch.floor(draws["luma_i"][:, 0] * 2), zero)
    c = (eye4[None] - 2 * vv[None] * i[:, None, None]) @ c
    if channels > 1:
        th = torch.where(gate("hue_g"), (draws["hue_t"][:, 0] * 2 - 1) * math.pi * hue_max, zero)
        cs, sn = torch.cos(th), torch.sin(th)
        cc = 1 - cs
        vx, vy, vz = v[0], v[1], v[2]
        m = _eye(b, 4)
        m[:, 0, 0] = vx * vx * cc + cs; m[:, 0, 1] = vx * vy * cc - vz * sn; m[:, 0, 2] = vx * vz * cc + vy * sn
        m[:, 1, 0] = vy * vx * cc + vz * sn; m[:, 1, 1] = vy * vy * cc + cs; m[:, 1, 2] = vy * vz * cc - vx * sn
        m[:, 2, 0] = vz * vx * cc - vy * sn; m[:, 2, 1] = vz * vy * cc + vx * sn; m[:, 2, 2] = vz * vz * cc + cs
        c = m @ c
        s = torch.where(gate("sat_g"), torch.exp2(draws["sat_s"][:, 0] * saturation_std), one)
        c = (vv[None] + (eye4 - vv)[None] * s[:, None, None]) @ c
    return c


def margins(g_inv, width, height, hz_pad=3):
    """Reflection padding (mx0, my0, mx1, my1) that keeps the transformed image inside the source."""
    cx, cy = (width - 1) / 2, (height - 1) / 2
    cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float64)
    cp = g_inv @ cp.t()  # [B, xyz, idx]
    m = cp[:, :2, :].permute(1, 0, 2).flatten(1)  # [xy, B*idx]
    m = torch.cat([-m, m]).max(dim=1).values  # x0, y0, x1, y1
    m = m + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=torch.float64)
    m = torch.maximum(m, torch.zeros(4, dtype=torch.float64))
    m = torch.minimum(m, torch.tensor([width - 1, height - 1] * 2, dtype=torch.float64))
    return [int(v) for v in torch.ceil(m).tolist()]


def _fir(x, f, axis):
    c = x.shape[1]
    w = f.view(1, 1, -1, 1) if axis == 2 else f.view(1, 1, 1, -1)
    return F.conv2d(x, w.repeat(c, 1, 1, 1), groups=c)


def upsample2x(x, f):
    """Zero insertion, pad (6, 5), true convolution with f, gain 4 (2 per axis): 2H x 2W out."""
    b, c, h, w = x.shape
    z = torch.zeros((b, c, h * 2, w * 2), dtype=x.dtype)
    z[:, :, ::2, ::2] = x
    z = F.pad(z, [6, 5, 6, 5])
    ff = (f * 2).flip(0)
    return _fir(_fir(z, ff, 2), ff, 3)


def downsample2x_crop(x, f, hz_pad=3):
    """Crop (hz_pad*2 - 5) = 1 per side, correlation with f, every second sample."""
    x = x[:, :, 1:-1, 1:-1]
    return _fir(_fir(x, f, 2), f, 3)[:, :, ::2, ::2]


def apply_geometry(images, g_inv):
    """images [B,C,H,W] float64/32, g_inv [B,3,3] (float64)."""
    b, c, h, w = images.shape
    f = torch.tensor(SYM6, dtype=images.dtype)
    f = f / f.sum()
    hz_pad = len(SYM6) // 4
    mx0, my0, mx1, my1 = margins(g_inv, w, h, hz_pad)
    x = F.pad(images, [mx0, mx1, my0, my1], mode="reflect")
    g = translate2d(b, (mx0 - mx1) / 2, (my0 - my1) / 2) @ g_inv
    x = upsample2x(x, f)
    g = scale2d(b, 2, 2) @ g @ scale2d(b, 0.5, 0.5)
    g = translate2d(b, -0.5, -0.5) @ g @ translate2d(b, 0.5, 0.5)
    shape = [b, c, (h + hz_pad * 2) * 2, (w + hz_pad * 2) * 2]
    g = scale2d(b, 2 / x.shape[3], 2 / x.shape[2]) @ g @ scale2d(b, shape[3] / 2, shape[2] / 2)
    grid = F.affine_grid(g[:, :2, :].to(images.dtype), shape, align_corners=False)
    x = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return downsample2x_crop(x, f, hz_pad)


def apply_colour(images, cm):
    b, c, h, w = images.shape
    cm = cm.to(images.dtype)
    x = images.reshape(b, c, h * w)
    if c == 3:
        x = cm[:, :3, :3] @ x + cm[:, :3, 3:]
    elif c == 1:
        m = cm[:, :3, :].mean(dim=1, keepdim=True)
        x = x * m[:, :, :3].sum(dim=2, keepdim=True) + m[:, :, 3:]
    else:
        raise ValueError("1 or 3 channels")
    return x.reshape(b, c, h, w)


def augment(images, p, draws):
    """The whole pipe for one batch at probability p with explicit random draws."""
    b, c, h, w = images.shape
    g_inv = geometry_matrix(draws, p, w, h)
    x = images
    if not torch.equal(g_inv, _eye(b, 3)):
        x = apply_geometry(x, g_inv)
    cm = colour_matrix(draws, p, c)
    if not torch.equal(cm, _eye(b, 4)):
        x = apply_colour(x, cm)
    return x
