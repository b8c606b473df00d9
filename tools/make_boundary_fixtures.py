#!/usr/bin/env python3
"""Fixtures for the two text / file formats on the drop-in boundary, produced by running the
REFERENCE's own ``Logger`` and ``model_checkpoint`` (src/core/evaluation.py:227-308) in the
build container:

* ``tests/golden/logger_lines.json``      -- scripted tracker values -> the exact printed lines;
* ``tests/golden/checkpoint_layout.json`` -- the key tree of a ``<step>.tar`` written by the
  reference for small networks after one ``torch.optim.Adam`` step: every key, container type,
  tensor shape and dtype (no weights: the layout is the contract).

The reference module imports ``torchvision`` and ``cleanfid`` at the top although neither
function touches them; both are absent offline, so empty stand-in modules are placed in
``sys.modules`` (next to the ``ada`` / ``tomllib`` stand-ins of tools/make_golden.py).  Nothing
of the reference is copied: only its OUTPUTS are stored.

Usage:  python tools/make_boundary_fixtures.py
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
import types
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from tools.make_golden import reference_ns  # noqa: E402

# scripted per-step values for the Logger (two print intervals, awkward magnitudes included)
LOGGER_SCRIPT = [
    {"steps": 150000, "at": 500, "values": {
        "log_total_disc_losses": [0.25, 0.2431, 0.26001], "log_disc_real_accs": [0.1, -0.05, 0.33333],
        "log_disc_fake_accs": [-0.2, 0.0, 0.125], "log_total_gen_losses": [12.5, 11.75, 10.123456789],
        "log_gan_losses": [0.9, 0.8, 0.7], "log_idt_losses": [0.41, 0.39, 0.4], "log_rec_losses": [0.5, 0.52, 0.48],
        "log_kl_losses": [1234.5678, 1000.0, 999.9], "log_path_losses": [1e-5, 2e-5, 3e-5],
        "log_style_losses": [1.0, 1.0, 1.0], "log_ada_ps": [0.0, 0.0, 0.131072]}},
    {"steps": 150000, "at": 150000, "values": {
        "log_total_disc_losses": [1e-7], "log_disc_real_accs": [1.0], "log_disc_fake_accs": [-1.0],
        "log_total_gen_losses": [123456.789], "log_gan_losses": [0.333333333], "log_idt_losses": [2.5e-3],
        "log_rec_losses": [0.1], "log_kl_losses": [5e8], "log_path_losses": [0.0], "log_style_losses": [0.2],
        "log_ada_ps": [0.262144]}},
]

NET_ARGS = {"G": dict(input_nc=1, w_dim=6, image_size=(32, 32), min_latent_resolution=8, n_resnet_blocks=3,
                      start_filters=8),
            "D": dict(input_nc=1), "S": dict(input_nc=1, w_dim=6),
            "M": dict(features=6, n_layers=2, style_mixing_prob=0.9)}


def layout(obj):
    """JSON-able description of a checkpoint value: containers keep their keys, tensors become
    {"tensor": shape, "dtype": ...}, scalars their type name."""
    if isinstance(obj, torch.Tensor):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {"dict": {str(k): layout(v) for k, v in obj.items()}}
    if isinstance(obj, (list, tuple)):
        return {type(obj).__name__: [layout(v) for v in obj]}
    return type(obj).__name__


def reference_evaluation():
    reference_ns()  # ada / tomllib stand-ins + sys.path
    for name in ("torchvision", "cleanfid", "cleanfid.fid"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cleanfid"].fid = sys.modules["cleanfid.fid"]
    from src.core import evaluation as rev

    return rev


def main():
    rev = reference_evaluation()
    from src.core.training import ImageBuffer
    from src.model import builder as rbd
    from src.model.loss import ADAp

    out = Path(ROOT) / "tests" / "golden"
    lines = []
    for item in LOGGER_SCRIPT:
        lg = rev.Logger(item["steps"])
        for k, v in item["values"].items():
            getattr(lg, k).extend(v)
        lines.append({"steps": item["steps"], "at": item["at"], "values": item["values"],
                      "line": lg.print(item["at"])})
        assert all(getattr(lg, k) == [] for k in item["values"])  # print() resets
    (out / "logger_lines.json").write_text(json.dumps(lines, indent=1) + "\n")

    torch.manual_seed(0)
    nets = {"G": rbd.Generator(**NET_ARGS["G"]), "D": rbd.Discriminator(**NET_ARGS["D"]),
            "S": rbd.StyleExtractor(**NET_ARGS["S"]), "M": rbd.MappingNetwork(**NET_ARGS["M"])}
    opts = {k: torch.optim.Adam(n.parameters(), lr=2e-3, betas=(0.5, 0.99)) for k, n in nets.items()}
    for n, o in zip(nets.values(), opts.values()):  # one Adam step so the optimiser state exists
        for p in n.parameters():
            p.grad = torch.full_like(p, 0.5)
        o.step()
    buf = ImageBuffer(5)
    buf(torch.zeros(3, 1, 32, 32))
    ada_p = ADAp(256, 5.12e-4, 4, 0.6)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = {"training": {"checkpoint_directory": Path(tmp), "training_run": "r"}}
        rev.model_checkpoint(6, cfg, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"], opts["M"],
                             opts["S"], ada_p, buf)
        files = sorted(p.relative_to(tmp).as_posix() for p in Path(tmp).rglob("*") if p.is_file())
        blob = torch.load(Path(tmp) / files[0], map_location="cpu", weights_only=True)
    desc = {"files": files, "net_args": {k: {a: (list(v) if isinstance(v, tuple) else v) for a, v in kw.items()}
                                          for k, kw in NET_ARGS.items()},
            "layout": layout(blob)}
    (out / "checkpoint_layout.json").write_text(json.dumps(desc, indent=1) + "\n")
    print("wrote logger_lines.json, checkpoint_layout.json:", files, sorted(blob))


if __name__ == "__main__":
    main()
