"""Implementation namespaces for tests/cases.py: uniform constructors over the CPU oracle
and the HIP product path.  (The reference namespace lives in tools/make_golden.py: the
reference exists only in the build container.)"""

from __future__ import annotations

from types import SimpleNamespace

import torch


def oracle_ns():
    from oracle import model as om
    from oracle import training as ot

    return SimpleNamespace(
        name="oracle",
        conv=lambda cin, cout, k, pad, bias: om.EqConv(cin, cout, k, pad, use_bias=bias),
        modconv=lambda cin, cout, k, wdim, pad: om.ModConv(cin, cout, k, wdim, pad),
        up=om.Up, down=om.Down,
        smooth=lambda: _OracleBlur(),
        resblock=om.ResBlock, modresblock=om.ModResBlock,
        Generator=om.Generator, Discriminator=om.Discriminator,
        StyleExtractor=om.StyleExtractor, MappingNetwork=om.MappingNetwork,
        style_cycle_loss_func=ot.style_cycle_loss_func, kl_loss_func=ot.kl_loss_func,
        path_loss_func=ot.path_loss_func, ADAp=ot.ADAp, ImageBuffer=ot.ImageBuffer,
        discriminator_step=ot.discriminator_step, generator_step=ot.generator_step,
        make_ada=ot.IdentityADA,
    )


class _OracleBlur(torch.nn.Module):
    def forward(self, x):
        from oracle.model import f_blur

        return f_blur(x)


def product_ns(precision="fp32"):
    """HIP path.  ``precision``: "fp32" (bf16x3 split MFMA, parity mode), "bf16", or "fp8" (config #5)."""
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd.core import training as pt
    from one_to_many_gan_amd.model import blocks as pb
    from one_to_many_gan_amd.model import builder as pbd
    from one_to_many_gan_amd.model import layers as pl
    from one_to_many_gan_amd.model import loss as plo

    pk.set_precision(precision)
    pk.ops.FP8_EVERYWHERE = precision == "fp8"  # the parity cases are far below the sizes fp8 is used at by default
    return SimpleNamespace(
        name=f"hip-{precision}",
        conv=lambda cin, cout, k, pad, bias: pl.EqualisedConv2d(cin, cout, k, padding=pad, use_bias=bias),
        modconv=lambda cin, cout, k, wdim, pad: pl.Conv2dWeightModulate(cin, cout, k, wdim, pad),
        conv_reflect=lambda m, p: pl.ReflectFused(m, p),
        modconv_reflect=lambda m, p: pl.ReflectFused(m, p),
        up=pl.UpSample, down=pl.DownSample, smooth=pl.Smooth,
        resblock=pb.ResnetBlock, modresblock=pb.ModulatedResnetBlock,
        Generator=pbd.Generator, Discriminator=pbd.Discriminator,
        StyleExtractor=pbd.StyleExtractor, MappingNetwork=pbd.MappingNetwork,
        style_cycle_loss_func=plo.style_cycle_loss_func, kl_loss_func=plo.kl_loss_func,
        path_loss_func=plo.path_loss_func, ADAp=plo.ADAp, ImageBuffer=pt.ImageBuffer,
        discriminator_step=pt.discriminator_step, generator_step=pt.generator_step,
        make_ada=pk.IdentityADA,
        make_adam=pk.make_adam,
    )
