"""Adaptive discriminator augmentation: product vs the oracle's
restatement of the published pipe, driven by the same random draws.  Parity against the
un-vendored pytorch-ada itself is unpinned (oracle/ada.py header)."""

import numpy as np
import pytest
import torch

from oracle import ada as O

SWITCHES = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1,
                lumaflip=1, hue=1, saturation=1)  # reference train.py:175-188


def _pipe(seed, p):
    from one_to_many_gan_amd import ada as P

    aug = P.AdaptiveDiscriminatorAugmentation(**SWITCHES, generator=torch.Generator().manual_seed(seed))
    aug.set_p(p)
    return aug


@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("p", [0.2, 0.95])
def test_host_matrices_match_oracle(channels, p):
    aug = _pipe(11, p)
    d = aug.draw(16)
    od = O.make_draws(16, torch.Generator().manual_seed(11))
    assert list(d) == [n for n, _, _ in O.DRAWS]
    assert all(np.array_equal(d[k], od[k].numpy()) for k in d)
    g, og = aug.geometry_matrix(d, 256, 128), O.geometry_matrix(od, p, 256, 128)
    c, oc = aug.colour_matrix(d, channels), O.colour_matrix(od, p, channels)
    assert np.abs(g - og.numpy()).max() < 1e-12 and np.abs(c - oc.numpy()).max() < 1e-12
    assert aug.margins(g, 256, 128) == O.margins(og, 256, 128)


def test_banded_operators_are_the_oracle_stages():
    from one_to_many_gan_amd import ada as P
    from one_to_many_gan_amd import resample as R

    f = torch.tensor(O.SYM6, dtype=torch.float64)
    f = f / f.sum()
    x = torch.randn(1, 1, 12, 18, dtype=torch.float64)
    up = O.upsample2x(torch.nn.functional.pad(x, [5, 2, 0, 7], mode="reflect"), f)
    ay, ax = torch.from_numpy(P._pad_up_operator(12, 0, 7)), torch.from_numpy(P._pad_up_operator(18, 5, 2))
    assert float((ay @ x[0, 0] @ ax.T - up[0, 0]).abs().max()) < 1e-12
    z = torch.randn(1, 1, 2 * 12 + 12, 2 * 18 + 12, dtype=torch.float64)
    dy, dx = torch.from_numpy(P._down_operator(12)), torch.from_numpy(P._down_operator(18))
    assert float((dy @ z[0, 0] @ dx.T - O.downsample2x_crop(z, f)[0, 0]).abs().max()) < 1e-12
    # what the device kernels are instantiated for
    assert R.banded(P._pad_up_operator(256, 100, 3))[2] <= 8
    assert R.banded(P._up_operator(300).T)[2] == 12 and R.banded(P._down_operator(256).T)[2] == 6


def test_p_zero_is_the_identity_without_touching_the_gpu():
    aug = _pipe(0, 0.0)
    x = torch.randn(2, 3, 8, 8)
    assert aug(x) is x


def test_oracle_geometry_properties():
    """The restated pipe: identity reconstructs, integer shifts are exact away from the border."""
    yy, xx = torch.meshgrid(torch.arange(32.0, dtype=torch.float64), torch.arange(32.0, dtype=torch.float64), indexing="ij")
    img = torch.stack([torch.sin(xx / 5) + torch.cos(yy / 7), torch.sin((xx + yy) / 9), torch.cos(xx / 4)])[None]
    assert float((O.apply_geometry(img, O._eye(1, 3)) - img).abs().max()) < 1e-10
    out = O.apply_geometry(img, O.translate2d(1, -3.0, 2.0))
    assert float((out[:, :, 4:-4, 6:-6] - img[:, :, 6:-2, 3:-9]).abs().max()) < 1e-10


def _smooth_images(b, c, h, w):
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    base = [torch.sin(xx / 5 + k) + torch.cos(yy / 7 - k) * 0.5 for k in range(c)]
    g = torch.Generator().manual_seed(3)
    return torch.stack([torch.stack(base) * (0.6 + 0.1 * i) for i in range(b)]) + \
        0.05 * torch.randn(b, c, h, w, generator=g, dtype=torch.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_pipe_matches_oracle_forward_and_backward(channels, precision, tol):
    import one_to_many_gan_amd as o2m

    o2m.set_precision(precision)
    try:
        b, h, w = 4, 40, 56
        img = _smooth_images(b, channels, h, w)
        probe = torch.randn(b, channels, h, w, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
        for seed in (1, 2, 3):
            aug = _pipe(seed, 0.85)
            d = aug.draw(b)
            od = {k: torch.from_numpy(v) for k, v in d.items()}
            xo = img.clone().requires_grad_(True)
            yo = O.augment(xo, 0.85, od)
            (yo * probe).sum().backward()
            xp = img.float().cuda().requires_grad_(True)
            yp = aug(xp, draws=d)
            (yp.float() * probe.float().cuda()).sum().backward()
            scale = float(yo.detach().abs().max())
            assert float((yp.detach().double().cpu() - yo.detach()).abs().max()) < tol * scale, seed
            gscale = float(xo.grad.abs().max())
            assert float((xp.grad.double().cpu() - xo.grad).abs().max()) < tol * gscale, seed
    finally:
        o2m.set_precision("bf16")


@pytest.mark.gpu
def test_ada_kernels_against_torch():
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(0)
    B, C, Cp = 3, 3, 8
    # reflect fold = adjoint of F.pad(reflect) with asymmetric margins
    gp = torch.zeros(B, 19 + 5 + 0, 23 + 7 + 22, Cp, device="cuda")
    gp[..., :C] = torch.randn(B, 24, 52, C, device="cuda")
    gx = torch.empty(B, 19, 23, Cp, device="cuda")
    H.reflect_fold(gp, gx, 5, 7)
    x = torch.zeros(B, C, 19, 23, dtype=torch.float64, requires_grad=True)
    (F.pad(x, [7, 22, 5, 0], mode="reflect") * gp[..., :C].double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert float((gx[..., :C].double().cpu().permute(0, 3, 1, 2) - x.grad).abs().max()) < 1e-5
    # bilinear affine resampling and its adjoint
    src = torch.zeros(B, 30, 44, Cp, device="cuda")
    src[..., :C] = torch.randn(B, 30, 44, C, device="cuda")
    theta = (torch.eye(2, 3)[None] + 0.3 * torch.randn(B, 2, 3)).float()
    out = torch.empty(B, 26, 34, Cp, device="cuda")
    H.ada_grid_sample(src, theta.reshape(B, 6).cuda(), out)
    s64 = src[..., :C].double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.grid_sample(s64, F.affine_grid(theta.double(), [B, C, 26, 34], align_corners=False), mode="bilinear",
                        padding_mode="zeros", align_corners=False)
    assert float((out[..., :C].double().cpu().permute(0, 3, 1, 2) - ref.detach()).abs().max()) < 1e-4
    gy = torch.zeros(B, 26, 34, Cp, device="cuda")
    gy[..., :C] = torch.randn(B, 26, 34, C, device="cuda")
    gsrc = torch.full((B, 30, 44, Cp), 9.0, device="cuda")  # fully overwritten
    H.ada_grid_sample_bwd(gy, theta.reshape(B, 6).cuda(), gsrc)
    (ref * gy[..., :C].double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert float((gsrc[..., :C].double().cpu().permute(0, 3, 1, 2) - s64.grad).abs().max()) < 1e-3
    assert float(gsrc[..., C:].abs().max()) == 0.0
    # colour affine
    m = torch.randn(B, 3, 4)
    y = torch.empty_like(src)
    H.ada_colour(src, m.cuda(), y, C)
    want = torch.einsum("bck,bhwk->bhwc", m[:, :, :3].double(), src[..., :C].double().cpu()) + m[:, None, None, :, 3].double()
    assert float((y[..., :C].double().cpu() - want).abs().max()) < 1e-5 and float(y[..., C:].abs().max()) == 0.0


@pytest.mark.gpu
def test_steps_run_with_augmentation_active():
    """discriminator_step / generator_step with the augmentation at p = 0.6 (training.py:100,104,200):
    finite losses, the generator receives gradient through the augmented fake images."""
    import one_to_many_gan_amd as o2m
    import train
    from one_to_many_gan_amd.core.training import ImageBuffer, discriminator_step, generator_step
    from one_to_many_gan_amd.model.loss import ADAp
    from tests.cases import make_config

    o2m.set_precision("bf16")
    cfg = make_config(1, (64, 64), 4)
    dev = torch.device("cuda:0")
    nets, opts = train.build(cfg, dev)
    ada = o2m.AdaptiveDiscriminatorAugmentation(**o2m.REFERENCE_ADA_SWITCHES,
                                                generator=torch.Generator().manual_seed(0)).to(dev)
    ada.set_p(0.6)
    ada_p = ADAp(ada_e=cfg["ada"]["ada_overfitting_measurement_n_images"], ada_adjustment_size=cfg["ada"]["ada_adjustment_size"],
                 batch_size=4, discriminator_overfitting_target=cfg["ada"]["discriminator_real_acc_target"])
    prints, marks = train.synthetic_batches(1, cfg, dev), train.synthetic_batches(2, cfg, dev)
    buf = ImageBuffer(cfg["training"]["image_buffer_size"])
    before = nets["G"].decoder[-2].weight.weight.detach().clone()
    for _ in range(2):
        d_loss, _ = discriminator_step(cfg, dev, nets["D"], nets["G"], nets["M"], opts["D"], prints, marks, buf, ada, ada_p)
        g_loss, parts = generator_step(cfg, dev, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["M"], opts["S"],
                                       prints, marks, ada)
        assert all(np.isfinite(v) for v in (d_loss, g_loss, *parts))
    assert not torch.equal(before, nets["G"].decoder[-2].weight.weight.detach())


@pytest.mark.parametrize("n,m0,m1", [(256, 107, 110), (40, 0, 0), (33, 5, 31), (64, 63, 63), (8, 7, 0)])
def test_vectorised_operator_construction_matches_dense(n, m0, m1):
    """banded_coo (COO triplets, vectorised) == banded (dense matrix, row loops) for the per-call
    operators of the pipe, including margins as large as the image."""
    from one_to_many_gan_amd import ada as P
    from one_to_many_gan_amd import resample as R

    op = P._pad_up_coo(n, m0, m1)
    s1, w1, t1 = R.banded(P._dense(op))
    s2, w2, t2 = R.banded_coo(*op)
    assert t1 == t2 and np.array_equal(s1, s2) and np.allclose(w1, w2, atol=1e-7)
    L = n + m0 + m1
    opt = P._up_t_coo(L)
    assert np.allclose(P._dense(opt), P._up_operator(L).T)
    s1, w1, t1 = R.banded(P._dense(opt))
    s2, w2, t2 = R.banded_coo(*opt)
    assert t1 == t2 and np.array_equal(s1, s2) and np.allclose(w1, w2, atol=1e-7)
    # interior rows sum to 2 x (one polyphase of the normalised filter) = 1: constants are preserved
    # (the first / last rows lose taps to the zero extension of the upsampler, as in the published pipe)
    rows = P._dense(op).sum(axis=1)
    assert np.allclose(rows[6:-6], 1.0, atol=1e-6)


def test_margins_stay_inside_the_image_for_extreme_transforms():
    from one_to_many_gan_amd import ada as P

    aug = _pipe(3, 1.0)
    for seed in range(20):
        aug.generator.manual_seed(seed)
        g = aug.geometry_matrix(aug.draw(8), 64, 48)
        mx0, my0, mx1, my1 = aug.margins(g, 64, 48)
        assert 0 <= mx0 <= 63 and 0 <= mx1 <= 63 and 0 <= my0 <= 47 and 0 <= my1 <= 47
