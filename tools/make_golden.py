#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running tests/cases.py on the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  The reference is imported as-is; the only stand-ins placed in ``sys.modules`` are
(SURVEY.md Appendix C): ``tomllib`` := ``tomli`` (container Python is 3.10) and an ``ada``
module whose ``AdaptiveDiscriminatorAugmentation`` is an identity ``nn.Module`` with
``set_p`` -- pytorch-ada @ 99754cb4 is an un-vendored VCS dependency that cannot be
fetched offline, and every parity configuration holds augmentation at p = 0.

Usage:  python tools/make_golden.py [case ...]     (default: all cases)
"""

from __future__ import annotations

import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def reference_ns():
    import tomli

    sys.modules.setdefault("tomllib", tomli)
    ada_mod = types.ModuleType("ada")

    class AdaptiveDiscriminatorAugmentation(torch.nn.Module):
        def __init__(self, **_kw):
            super().__init__()
            self.p = 0.0

        def set_p(self, p):
            self.p = float(p)

        def forward(self, x):
            return x

    ada_mod.AdaptiveDiscriminatorAugmentation = AdaptiveDiscriminatorAugmentation
    sys.modules["ada"] = ada_mod
    sys.path.insert(0, REF)
    from src.core import training as rt
    from src.model import blocks as rb
    from src.model import builder as rbd
    from src.model import layers as rl
    from src.model import loss as rlo

    return SimpleNamespace(
        name="reference",
        conv=lambda cin, cout, k, pad, bias: rl.EqualisedConv2d(cin, cout, k, padding=pad, use_bias=bias),
        modconv=lambda cin, cout, k, wdim, pad: rl.Conv2dWeightModulate(cin, cout, k, wdim, pad),
        up=rl.UpSample, down=rl.DownSample, smooth=rl.Smooth,
        resblock=rb.ResnetBlock, modresblock=rb.ModulatedResnetBlock,
        Generator=rbd.Generator, Discriminator=rbd.Discriminator,
        StyleExtractor=rbd.StyleExtractor, MappingNetwork=rbd.MappingNetwork,
        style_cycle_loss_func=rlo.style_cycle_loss_func, kl_loss_func=rlo.kl_loss_func,
        path_loss_func=rlo.path_loss_func, ADAp=rlo.ADAp, ImageBuffer=rt.ImageBuffer,
        discriminator_step=rt.discriminator_step, generator_step=rt.generator_step,
        make_ada=AdaptiveDiscriminatorAugmentation,
    )


def main(argv):
    from tests.cases import CASES, run_case

    torch.set_num_threads(os.cpu_count() or 1)
    ns = reference_ns()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    names = argv or list(CASES)
    for name in names:
        t0 = time.time()
        res = run_case(name, ns, "cpu")
        arrays = {k: v.numpy().astype(np.float32) for k, v in res.items()}
        path = os.path.join(out_dir, f"{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{name:24s} {len(arrays):4d} arrays {os.path.getsize(path) / 1024:8.1f} KiB "
              f"{time.time() - t0:6.1f}s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
