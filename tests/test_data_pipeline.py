"""Input pipeline: host logic on CPU, the gather kernel on the GPU."""

import itertools

import numpy as np
import pytest
import torch

from oracle import data as odata


def _make_folder(tmp_path, n=7, size=(40, 24), gray=False):
    from PIL import Image

    root = tmp_path / "shoes" / "train"
    root.mkdir(parents=True)
    rng = np.random.default_rng(0)
    for i in range(n):
        a = rng.integers(0, 256, size=(size[0], size[1]) if gray else (size[0], size[1], 3), dtype=np.uint8)
        Image.fromarray(a).save(root / f"im{i}.{'png' if i % 2 else 'jpg'}")
    return tmp_path / "shoes"


def test_transforms_match_the_published_definitions(tmp_path):
    from PIL import Image

    from one_to_many_gan_amd.data import datasets as D

    a = np.random.default_rng(1).integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    t = D.Compose([D.ToTensor(), D.Normalize((0.5,), (0.5,))])(Image.fromarray(a))
    ref = odata.normalize(odata.to_tensor(torch.from_numpy(a)))
    assert t.shape == (3, 9, 11) and torch.equal(t, ref)
    g = D.ToTensor()(Image.fromarray(a[:, :, 0]))
    assert g.shape == (1, 9, 11)
    assert D.Resize((5, 7))(Image.fromarray(a)).size == (7, 5)  # PIL size is (w, h)
    assert D.Resize(6)(Image.fromarray(a)).size == (7, 6)  # shorter edge (h = 9) -> 6


def test_shoe_dataset_mirrors_reference_contract(tmp_path):
    from one_to_many_gan_amd.data import datasets as D

    root = _make_folder(tmp_path)
    tf = D.Compose([D.Resize((32, 16)), D.ToTensor(), D.Normalize((0.5,), (0.5,))])
    ds = D.ShoeDataset(root, mode="train", transform=tf)
    assert len(ds) == 7 and ds[0].shape == (3, 32, 16) and ds[0].dtype == torch.float32
    assert float(ds[0].min()) >= -1.0 and float(ds[0].max()) <= 1.0
    # flip_prob: 0 never flips, 1 always does (RandomHorizontalFlip semantics)
    never = D.ShoeDataset(root, mode="train", transform=tf, flip_prob=0.0)
    always = D.ShoeDataset(root, mode="train", transform=tf, flip_prob=1.0)
    assert torch.equal(never[3], never.images[3]) and torch.equal(always[3], always.images[3].flip(-1))
    # the uint8 images kept for the device pool reproduce the float ones exactly
    assert torch.equal(odata.normalize(odata.to_tensor(torch.from_numpy(ds.raw[2]))), ds.images[2])
    (tmp_path / "empty" / "train").mkdir(parents=True)
    with pytest.raises(FileNotFoundError):
        D.ShoeDataset(tmp_path / "empty", mode="train", transform=tf)


def test_loader_plan_is_the_reference_sampler():
    """shuffle=True draws torch.randperm(N, generator=g) per epoch like RandomSampler; drop_last."""
    from one_to_many_gan_amd.data import datasets as D

    class Pool:
        def __len__(self):
            return 23

    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    loader = D.DeviceLoader(Pool(), 4, generator=g1, flip_prob=0.0)  # flip_prob 0 still draws: keep g aligned
    plan = loader.plan()
    sampler = list(torch.utils.data.RandomSampler(range(23), generator=g2))
    assert len(loader) == 5 and len(plan) == 5
    assert torch.cat([p[0] for p in plan]).tolist() == sampler[:20]
    assert all(int(p[1].sum()) == 0 for p in plan)
    keep = D.DeviceLoader(Pool(), 4, drop_last=False, shuffle=False)
    assert len(keep) == 6 and keep.plan()[-1][0].tolist() == [20, 21, 22]
    with pytest.raises(ValueError):
        D.DeviceLoader(Pool(), 0)


def test_pool_refuses_cpu_and_foreign_transforms(tmp_path):
    from one_to_many_gan_amd.data import datasets as D

    with pytest.raises(RuntimeError):
        D.DeviceImagePool(torch.zeros(2, 4, 4, 3, dtype=torch.uint8), "cpu")
    root = _make_folder(tmp_path, n=2)
    ds = D.ShoeDataset(root, mode="train", transform=D.Compose([D.ToTensor()]))
    with pytest.raises(ValueError):
        D.DeviceImagePool(ds, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_gather_images_matches_oracle(channels):
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(0)
    N, B, Hh, Ww = 11, 6, 20, 300  # W > one block of 256 threads
    pool = torch.randint(0, 256, (N, Hh, Ww, channels), dtype=torch.uint8)
    idx = torch.tensor([3, 10, 0, 3, 7, 9], dtype=torch.int32)
    flip = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.uint8)
    ref = odata.batch(pool, idx, flip)  # (B, C, H, W) fp32
    for dt in (torch.float32, torch.bfloat16):
        out = torch.full((B, Hh, Ww, 8), 7.0, dtype=dt, device="cuda")
        H.gather_images(pool.cuda(), idx.cuda(), flip.cuda(), out)
        got = out.float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(got[:, channels:], torch.zeros_like(got[:, channels:]))  # padding channels zeroed
        want = ref if dt == torch.float32 else ref.to(torch.bfloat16).float()
        assert torch.equal(got[:, :channels], want)  # bit-exact (fp32) / correctly rounded (bf16)


@pytest.mark.gpu
def test_device_loader_feeds_the_step_without_a_layout_pass(tmp_path):
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd import ops
    from one_to_many_gan_amd.data import datasets as D

    o2m.set_precision("fp32")
    try:
        root = _make_folder(tmp_path, n=9, size=(16, 16))
        tf = D.Compose([D.Resize((16, 16)), D.ToTensor(), D.Normalize((0.5,), (0.5,))])
        ds = D.ShoeDataset(root, mode="train", transform=tf)
        pool = D.DeviceImagePool(ds, "cuda")
        g = torch.Generator().manual_seed(1)
        loader = D.DeviceLoader(pool, 4, generator=g)
        plan = D.DeviceLoader(pool, 4, generator=torch.Generator().manual_seed(1)).plan()
        batches = list(loader)
        assert len(batches) == 2 and batches[0].shape == (4, 3, 16, 16)
        for (idx, flip), got in zip(plan, batches):
            want = torch.stack([ds.images[i].flip(-1) if f else ds.images[i] for i, f in zip(idx.tolist(), flip.tolist())])
            assert torch.equal(got.cpu(), want)  # the CPU dataset's tensors, bit for bit
            inner = ops.to_internal(got)
            assert inner.data_ptr() == got.data_ptr() and inner.shape == (4, 16, 16, 8)  # zero-copy
        it = itertools.cycle(loader)  # the reference's idiom works unchanged
        assert next(it).shape == (4, 3, 16, 16)
        first3 = list(itertools.islice(loader.cycle(), 3))
        assert len(first3) == 3
    finally:
        o2m.set_precision("bf16")


def test_every_image_is_served_once_per_epoch():
    from one_to_many_gan_amd.data import datasets as D

    class Pool:
        def __len__(self):
            return 37

    loader = D.DeviceLoader(Pool(), 5, generator=torch.Generator().manual_seed(9), drop_last=False)
    for _ in range(3):  # a fresh permutation per epoch
        idx = torch.cat([p[0] for p in loader.plan()]).tolist()
        assert sorted(idx) == list(range(37))
    dropped = D.DeviceLoader(Pool(), 5, generator=torch.Generator().manual_seed(9))
    idx = torch.cat([p[0] for p in dropped.plan()]).tolist()
    assert len(idx) == 35 and len(set(idx)) == 35
