"""CPU tests of the text / file formats on the drop-in boundary, against fixtures produced by the
reference's own ``Logger`` and ``model_checkpoint`` (tools/make_boundary_fixtures.py):

* ``Logger.print`` emits the reference's line byte for byte;
* ``model_checkpoint`` writes the reference's eleven keys with the same nested layout
  (``torch.optim.Adam`` state included), so the reference's tooling reads our files;
* ``load_checkpoint`` reads a file in the REFERENCE'S layout: weights, Adam moments and the image
  history pool all arrive (round 1 silently dropped the pool and the optimiser states).
"""

import json
import os
import warnings

import pytest
import torch

from tools.make_boundary_fixtures import layout

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REF_KEYS = ["generator_state_dict", "generator_optim_state_dict", "discriminator_state_dict",
            "discriminator_optim_state_dict", "mapping_network_state_dict", "mapping_network_optim_state_dict",
            "style_extractor_state_dict", "style_extractor_optim_state_dict", "ada_p", "image_buffer_images",
            "image_buffer_size"]


def _fixture():
    return json.load(open(os.path.join(GOLDEN, "checkpoint_layout.json")))


def _nets(mod, args):
    a = {k: dict(v) for k, v in args.items()}
    a["G"]["image_size"] = tuple(a["G"]["image_size"])
    return {"G": mod.Generator(**a["G"]), "D": mod.Discriminator(**a["D"]),
            "S": mod.StyleExtractor(**a["S"]), "M": mod.MappingNetwork(**a["M"])}


def test_logger_prints_the_reference_line():
    from one_to_many_gan_amd.core.evaluation import Logger

    for item in json.load(open(os.path.join(GOLDEN, "logger_lines.json"))):
        lg = Logger(item["steps"])
        for k, v in item["values"].items():
            getattr(lg, k).extend(v)
        assert lg.print(item["at"]) == item["line"]
        assert all(getattr(lg, k) == [] for k in item["values"])  # print() starts a new window


def test_checkpoint_has_the_reference_layout(tmp_path):
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.core.evaluation import model_checkpoint
    from one_to_many_gan_amd.core.training import ImageBuffer
    from one_to_many_gan_amd.model import builder as pb
    from one_to_many_gan_amd.model.loss import ADAp

    fx = _fixture()
    assert list(fx["layout"]["dict"]) == REF_KEYS
    nets = _nets(pb, fx["net_args"])
    opts = {k: o2m.make_adam(n, 2e-3, (0.5, 0.99)) for k, n in nets.items()}
    for o in opts.values():
        o.step_t.fill_(1.0)  # the fixture was taken after one optimiser step
    buf = ImageBuffer(5)
    buf(torch.zeros(3, 1, 32, 32))
    cfg = {"training": {"checkpoint_directory": tmp_path, "training_run": "r"}}
    path = model_checkpoint(6, cfg, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"], opts["M"],
                            opts["S"], ADAp(256, 5.12e-4, 4, 0.6), buf)
    assert [p.relative_to(tmp_path).as_posix() for p in tmp_path.rglob("*.tar")] == fx["files"] == ["r/models/7.tar"]
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert list(blob)[: len(REF_KEYS)] == REF_KEYS  # the reference's keys, in its order, then our extras
    assert set(blob) - set(REF_KEYS) == {"step", "ada_state"}
    ours = layout({k: blob[k] for k in REF_KEYS})
    assert ours == fx["layout"]
    # and torch's own Adam accepts the optimiser entries (what reference-side tooling would do)
    from oracle import model as om

    ref_nets = _nets(om, fx["net_args"])
    for k, name in (("G", "generator"), ("D", "discriminator"), ("M", "mapping_network"), ("S", "style_extractor")):
        ref_nets[k].load_state_dict(blob[f"{name}_state_dict"])
        torch.optim.Adam(ref_nets[k].parameters()).load_state_dict(blob[f"{name}_optim_state_dict"])


def test_reference_layout_checkpoint_loads(tmp_path):
    """A file in the reference's layout (oracle modules + torch.optim.Adam, keys as in the fixture)
    restores weights, Adam moments, step counters and the image pool."""
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.core.evaluation import load_checkpoint
    from one_to_many_gan_amd.core.training import ImageBuffer
    from one_to_many_gan_amd.model import builder as pb
    from one_to_many_gan_amd.model.loss import ADAp
    from oracle import model as om

    fx = _fixture()
    torch.manual_seed(3)
    ref = _nets(om, fx["net_args"])
    ref_opts = {k: torch.optim.Adam(n.parameters(), lr=2e-3, betas=(0.5, 0.99)) for k, n in ref.items()}
    for n, o in zip(ref.values(), ref_opts.values()):
        for _ in range(2):
            for p in n.parameters():
                p.grad = torch.randn_like(p)
            o.step()
    pool = [torch.randn(1, 1, 32, 32) for _ in range(4)]
    names = {"G": "generator", "D": "discriminator", "M": "mapping_network", "S": "style_extractor"}
    blob = {}
    for k, name in names.items():
        blob[f"{name}_state_dict"] = ref[k].state_dict()
        blob[f"{name}_optim_state_dict"] = ref_opts[k].state_dict()
    blob.update(ada_p=0.131072, image_buffer_images=pool, image_buffer_size=5)
    assert layout({k: blob[k] for k in REF_KEYS})["dict"].keys() == fx["layout"]["dict"].keys()
    path = tmp_path / "ref.tar"
    torch.save(blob, path)

    nets = _nets(pb, fx["net_args"])
    opts = {k: o2m.make_adam(n, 1e-3, (0.9, 0.999)) for k, n in nets.items()}
    buf, ada_p = ImageBuffer(5), ADAp(256, 5.12e-4, 4, 0.6)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # nothing may be missing from a reference file
        step = load_checkpoint(path, "cpu", nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                               opts["M"], opts["S"], ada_p, buf)
    assert step == 0  # the reference records no step
    assert buf.num_imgs == 4 and all(torch.equal(a, b) for a, b in zip(buf.images, pool))
    assert abs(ada_p() - 0.131072) < 1e-7
    for k in names:
        for (n1, p1), (n2, p2) in zip(nets[k].named_parameters(), ref[k].named_parameters()):
            assert n1 == n2 and torch.equal(p1, p2), (k, n1)
        o, ro = opts[k], ref_opts[k]
        assert float(o.step_t) == 2.0 and o.lr == 2e-3 and o.betas == (0.5, 0.99)
        for i, (p, off) in enumerate(zip(o.bucket.params, o.bucket.offsets)):
            st = ro.state[list(ref[k].parameters())[i]]
            assert torch.equal(o.exp_avg[off: off + p.numel()].view_as(p), st["exp_avg"])
            assert torch.equal(o.exp_avg_sq[off: off + p.numel()].view_as(p), st["exp_avg_sq"])


def test_incomplete_checkpoint_warns(tmp_path):
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.core.evaluation import load_checkpoint
    from one_to_many_gan_amd.core.training import ImageBuffer
    from one_to_many_gan_amd.model import builder as pb

    fx = _fixture()
    nets = _nets(pb, fx["net_args"])
    names = {"G": "generator", "D": "discriminator", "M": "mapping_network", "S": "style_extractor"}
    path = tmp_path / "weights_only.tar"
    torch.save({f"{n}_state_dict": nets[k].state_dict() for k, n in names.items()}, path)
    opt = o2m.make_adam(nets["G"], 1e-3)
    with pytest.warns(UserWarning) as rec:
        load_checkpoint(path, "cpu", nets["G"], nets["D"], nets["M"], nets["S"], generator_optimiser=opt,
                        image_buffer=ImageBuffer(5))
    text = " | ".join(str(w.message) for w in rec)
    assert "optimiser state" in text and "history pool" in text


def test_round1_flat_optimiser_state_still_loads():
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.model import builder as pb

    net = pb.MappingNetwork(6, 2, 0.9)
    opt = o2m.make_adam(net, 1e-3)
    n = opt.bucket.numel
    opt.load_state_dict({"step": torch.tensor([3.0]), "exp_avg": torch.arange(n, dtype=torch.float32),
                         "exp_avg_sq": torch.ones(n)})
    assert float(opt.step_t) == 3.0 and float(opt.exp_avg[5]) == 5.0
    sd = opt.state_dict()  # and comes back out in the torch layout
    assert sorted(sd) == ["param_groups", "state"] and float(sd["state"][0]["step"]) == 3.0
    opt2 = o2m.make_adam(pb.MappingNetwork(6, 2, 0.9), 1e-3)
    opt2.load_state_dict(sd)
    assert float(opt2.step_t) == 3.0
    for p, off in zip(opt.bucket.params, opt.bucket.offsets):  # (padding between slices is not state)
        assert torch.equal(opt2.exp_avg[off: off + p.numel()], opt.exp_avg[off: off + p.numel()])


def test_save_grid_writes_png(tmp_path):
    from one_to_many_gan_amd.core.evaluation import save_grid

    imgs = [[torch.rand(1, 8, 8) for _ in range(3)] for _ in range(2)]  # 2 columns x 3 rows
    save_grid(imgs, tmp_path / "g.png", (3, 2))
    assert (tmp_path / "g.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
