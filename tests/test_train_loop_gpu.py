"""GPU test of the training loop plumbing: train.run on synthetic data at BASELINE config #1's
shape (64x64x1, batch 4), checkpoint written with the reference's dictionary keys, resume."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_loop_checkpoint_and_resume(tmp_path):
    import train
    import one_to_many_gan_amd as o2m
    from tests.cases import make_config

    o2m.set_precision("bf16")
    cfg = make_config(1, (64, 64), 4)
    cfg["training"].update(checkpoint_directory=tmp_path, training_run="t", training_steps=3)
    cfg["evaluation"] = {"log_interval": 2, "checkpoint_interval": 2, "n_evaluation_images": 0,
                         "inference_batch_size": 4}
    dev = torch.device("cuda:0")
    lines = []
    nets, opts = train.run(cfg, dev, 3, train.synthetic_batches(1, cfg, dev), train.synthetic_batches(2, cfg, dev),
                           log=lines.append)
    # the reference's log line (evaluation.py:288-304), also appended to <run>/log (train.py:253-267)
    assert any(l.startswith("Step: 2/3, D loss: ") for l in lines) and any(l.startswith("Step: 3/3, ") for l in lines)
    logged = (tmp_path / "t" / "log").read_text().splitlines()
    assert len(logged) == 2 and logged[1].startswith("Step: 3/3, D loss: ") and logged[1].endswith(", ")
    ck = tmp_path / "t" / "models" / "3.tar"
    assert ck.exists() and (tmp_path / "t" / "models" / "2.tar").exists()
    # image grids of evaluation.py:122-221
    for name in ("translation_2.png", "decoding_2.png", "translation_3.png", "decoding_3.png"):
        assert (tmp_path / "t" / "images" / name).stat().st_size > 10000, name
    blob = torch.load(ck, map_location="cpu", weights_only=True)
    from tests.test_boundary_formats import REF_KEYS

    assert list(blob)[: len(REF_KEYS)] == REF_KEYS  # the reference's eleven keys, in its order
    assert len(blob["image_buffer_images"]) == 12  # 3 steps x batch 4 generated images pooled
    assert all(t.shape == (1, 1, 64, 64) and t.dtype == torch.float32 for t in blob["image_buffer_images"])
    assert float(blob["generator_optim_state_dict"]["state"][0]["step"]) == 3.0
    # resume continues from step 3 with identical weights and Adam state
    lines2 = []
    nets2, opts2 = train.run(cfg, dev, 4, train.synthetic_batches(1, cfg, dev), train.synthetic_batches(2, cfg, dev),
                             resume=ck, log=lines2.append)
    assert lines2[0].endswith("at step 3")
    assert float(opts2["G"].step_t) == 4.0
    # torch's own Adam reads the optimiser entry (reference-side tooling)
    # the oracle's modules load the same checkpoint (reference key layout)
    from oracle import model as om

    g = om.Generator(1, 6, (64, 64), 64, 7)
    g.load_state_dict(blob["generator_state_dict"])
    torch.optim.Adam(g.parameters()).load_state_dict(blob["generator_optim_state_dict"])


def test_train_loop_on_image_folders(tmp_path):
    """The reference's data path (train.py:118-169): image folders -> ShoeDataset -> HBM pool ->
    DeviceLoader.cycle() -> step functions, two steps at 64x64x1 with 6 + 5 PNG / JPG files."""
    import numpy as np
    from PIL import Image

    import train
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.data import datasets as D
    from tests.cases import make_config

    o2m.set_precision("bf16")
    rng = np.random.default_rng(3)
    for name, n in (("prints", 6), ("marks", 5)):
        d = tmp_path / name / "train"
        d.mkdir(parents=True)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, size=(80, 72), dtype=np.uint8)).save(d / f"{i}.{'png' if i % 2 else 'jpg'}")
    cfg = make_config(1, (64, 64), 4)
    cfg["training"].update(checkpoint_directory=tmp_path, training_run="f", training_steps=2)
    cfg["evaluation"] = {"log_interval": 1, "checkpoint_interval": 100, "n_evaluation_images": 0,
                         "inference_batch_size": 4}
    dev = torch.device("cuda:0")
    tf = D.Compose([D.Resize((64, 64)), D.ToTensor(), D.Normalize((0.5,), (0.5,))])
    g = torch.Generator().manual_seed(0)
    loaders = [D.DeviceLoader(D.DeviceImagePool(D.ShoeDataset(tmp_path / k, mode="train", transform=tf), dev), 4,
                              generator=g) for k in ("prints", "marks")]
    assert len(loaders[0]) == 1 and len(loaders[1]) == 1  # drop_last
    lines = []
    train.run(cfg, dev, 2, loaders[0].cycle(), loaders[1].cycle(), log=lines.append)
    assert any(l.startswith("Step: 2/2, ") for l in lines)
    assert all("nan" not in l.lower() for l in lines)
