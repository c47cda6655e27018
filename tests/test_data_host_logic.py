"""CPU: the data path.  The numpy oracle of SmallestMaxSize(INTER_AREA)+crop+flip+normalise is checked
against size-independent properties (cv2/albumentations are not installed: parity unpinned), then the host side
(e4t/data.py: plans, packing, sharding, prefetch loop) is run end to end through the op emulation."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

import image_prep_oracle as ipo
from test_unet_host_logic import emu_fp32  # noqa: F401


def _exact_area(img, nh, nw):
    """fractional-coverage box average in float64 (what INTER_AREA approximates in float32)"""
    def wmat(s, d):
        sc, m = s / d, np.zeros((d, s))
        for i in range(d):
            a, b = i * sc, (i + 1) * sc
            for j in range(int(np.floor(a)), min(int(np.ceil(b)), s)):
                m[i, j] = max(0.0, min(b, j + 1) - max(a, j))
            m[i] /= m[i].sum()
        return m
    return np.einsum("yh,hwc,xw->yxc", wmat(img.shape[0], nh), img.astype(np.float64), wmat(img.shape[1], nw))


def test_oracle_resize_properties():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (128, 192, 3), dtype=np.uint8)
    # 2x2: the (sum+2)>>2 rule
    want = (a.reshape(64, 2, 96, 2, 3).astype(int).sum((1, 3)) + 2) >> 2
    assert (ipo.resize_inter_area(a, 64, 96) == want).all()
    # integer 3x: box mean, nearest; PIL's BOX filter agrees to 1 LSB (different rounding)
    b = rng.integers(0, 256, (192, 192, 3), dtype=np.uint8)
    r = ipo.resize_inter_area(b, 64, 64)
    m = b.reshape(64, 3, 64, 3, 3).astype(np.float64).mean((1, 3))
    assert np.abs(r - m).max() <= 0.5 + 1e-4
    pil = np.asarray(Image.fromarray(b).resize((64, 64), Image.BOX)).astype(int)
    assert np.abs(r.astype(int) - pil).max() <= 1
    # general (non-integer) factors: within float32 rounding of the exact area average
    c = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    nh, nw = ipo.smallest_max_size_dims(97, 131, 64)
    assert (nh, nw) == (64, 86)
    r = ipo.resize_inter_area(c, nh, nw)
    assert np.abs(r - _exact_area(c, nh, nw)).max() <= 0.5 + 1e-3
    # constants stay constant on every branch (area, area-fast, enlarging)
    for (H, W, S) in ((97, 131, 64), (128, 192, 64), (40, 55, 64)):
        k = np.full((H, W, 3), 137, np.uint8)
        assert np.unique(ipo.resize_inter_area(k, *ipo.smallest_max_size_dims(H, W, S))).tolist() == [137]
    # enlarging: monotone ramps stay monotone and inside the source range
    ramp = np.repeat(np.arange(40, dtype=np.uint8)[:, None, None] * 5, 55, 1).repeat(3, 2)
    up = ipo.resize_inter_area(ramp, *ipo.smallest_max_size_dims(40, 55, 64)).astype(int)
    assert (np.diff(up[:, 0, 0]) >= 0).all() and up.min() >= 0 and up.max() <= 195
    # untouched when the short side already has the target size
    d = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    assert ipo.smallest_max_size_dims(64, 80, 64) == (64, 80)
    x = ipo.image_prep(d, 64, 0, 16, True)
    assert x.dtype == np.float32 and x.shape == (3, 64, 64)
    np.testing.assert_array_equal(x, (d[:, 16:80][:, ::-1] / 127.5 - 1.0).astype(np.float32).transpose(2, 0, 1))


def test_transform_plan_matches_oracle_dims_and_bounds():
    from e4t.data import make_transforms
    t = make_transforms(512, random_crop=True)
    rng = random.Random(3)
    for _ in range(200):
        h, w = rng.randint(200, 3000), rng.randint(200, 3000)
        nh, nw, y0, x0, flip = t.plan(h, w, rng)
        assert (nh, nw) == ipo.smallest_max_size_dims(h, w, 512)
        assert min(nh, nw) == 512 and 0 <= y0 <= nh - 512 and 0 <= x0 <= nw - 512 and flip in (0, 1)
    # half-way cases round to even like albumentations' py3round
    assert make_transforms(512).resized_dims(1024, 1025) == (512, 512)       # 512.5 -> 512
    assert make_transforms(512).resized_dims(1024, 1027) == (512, 514)       # 513.5 -> 514
    c = make_transforms(64, random_crop=False)
    assert c.plan(64, 100)[2:4] == (0, 18)


def _write_images(tmp_path, dims, seed=0):
    rng = np.random.default_rng(seed)
    sub = tmp_path / "imgs" / "nested"
    sub.mkdir(parents=True)
    imgs = {}
    for i, (h, w) in enumerate(dims):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        f = (sub if i % 2 else sub.parent) / f"im{i:02d}.png"
        Image.fromarray(a).save(f)
        imgs[str(f)] = a
    (sub.parent / "notes.txt").write_text("not an image")
    return str(tmp_path / "imgs"), imgs


def test_device_loader_end_to_end_cpu(tmp_path, emu_fp32):
    from e4t.data import DeviceLoader, E4TDataset
    dims = [(70, 90), (128, 128), (64, 64), (50, 200), (192, 130), (97, 131), (66, 64), (40, 44), (300, 80)]
    root, imgs = _write_images(tmp_path, dims)
    ds = E4TDataset(root, resolution=64)
    assert len(ds) == len(dims) and all(p.endswith(".png") for p in ds.dataset)
    ds.processor.random_crop, ds.processor.flip_p = False, 0.0          # deterministic plan -> comparable with the oracle
    seen = []
    for rank in range(2):
        ld = DeviceLoader(ds, batch_size=2, shuffle=True, num_workers=2, device="cpu", rank=rank, world=2, seed=5)
        assert len(ld) == 2
        order = [i for b in ld._indices() for i in b]
        seen += order
        k = 0
        for batch in ld:
            px = batch["pixel_values"]
            assert px.shape == (2, 3, 64, 64) and px.dtype == torch.float32
            for j in range(2):
                a = imgs[ds.dataset[order[k]]]
                nh, nw = ipo.smallest_max_size_dims(a.shape[0], a.shape[1], 64)
                want = ipo.image_prep(a, 64, (nh - 64) // 2, (nw - 64) // 2, False)
                np.testing.assert_array_equal(px[j].numpy(), want)
                k += 1
        assert k == 4
    assert len(set(seen)) == 8                                          # ranks see disjoint samples; the ragged tail is dropped
    # a second epoch reshuffles
    ld = DeviceLoader(ds, batch_size=2, shuffle=True, device="cpu", seed=5)
    e0 = [i for b in ld._indices() for i in b]
    list(ld)
    assert [i for b in ld._indices() for i in b] != e0


def test_pack_batch_rejects_bad_window():
    from e4t.data import pack_batch
    img = np.zeros((80, 100, 3), np.uint8)
    with pytest.raises(ValueError):
        pack_batch([dict(image=img, plan=(64, 80, 0, 17, 0))], 64)
    pool, table, total = pack_batch([dict(image=img, plan=(64, 80, 0, 16, 1)), dict(image=img[:70], plan=(64, 91, 0, 0, 0))], 64)
    assert table.tolist() == [[0, 80, 100, 64, 80, 0, 16, 1], [24000, 70, 100, 64, 91, 0, 0, 0]] and total == 24000 + 21008


def test_braceexpand_and_dataset_size(tmp_path):
    import json
    from e4t.data import braceexpand, get_dataset_size
    assert braceexpand("s/{000..002}.tar") == ["s/000.tar", "s/001.tar", "s/002.tar"]
    assert braceexpand("s/{8..10}.tar") == ["s/8.tar", "s/9.tar", "s/10.tar"]
    assert braceexpand("x{a,b}{1..2}") == ["xa1", "xa2", "xb1", "xb2"]
    assert braceexpand("plain.tar") == ["plain.tar"] and braceexpand("k{x}.tar") == ["k{x}.tar"]
    pat = str(tmp_path / "{00..01}.tar")
    (tmp_path / "00_stats.json").write_text(json.dumps({"successes": 7}))
    (tmp_path / "01_stats.json").write_text(json.dumps({"n_data": 5, "successes": 99}))
    assert get_dataset_size(pat) == (12, 2)
    (tmp_path / "sizes.json").write_text(json.dumps({"00.tar": 3, "01.tar": 4}))
    assert get_dataset_size(pat) == (7, 2)


def test_tar_shard_source_end_to_end_cpu(tmp_path, emu_fp32):
    import io
    import tarfile
    from e4t.data import DeviceLoader, TarShardDataset
    rng = np.random.default_rng(2)
    imgs = {}
    for s in range(2):
        with tarfile.open(tmp_path / f"shard-{s}.tar", "w") as tf:
            def add(name, data):
                ti = tarfile.TarInfo(name)
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))
            for i in range(5):
                a = rng.integers(0, 256, (70 + 9 * i, 90 + 5 * s, 3), dtype=np.uint8)
                b = io.BytesIO()
                Image.fromarray(a).save(b, format="PNG")          # lossless, stored under the "jpg" key like a webdataset shard
                key = f"{s}{i:04d}"
                add(f"{key}.jpg", b.getvalue())
                add(f"{key}.txt", b"a caption")
                imgs[a.shape[:2]] = a
            add(f"{s}9998.txt", b"caption without an image")                        # filtered: no jpg member
            add(f"{s}9999.jpg", b"this is not an image")                            # skipped with a warning at decode time
    ds = TarShardDataset(str(tmp_path / "shard-{0..1}.tar"), resolution=64, shuffle_buffer=4)
    assert len(ds.shards) == 2
    ds.processor.random_crop, ds.processor.flip_p = False, 0.0
    ld = DeviceLoader(ds, batch_size=3, num_workers=2, device="cpu", seed=1)
    with pytest.raises(TypeError):
        len(ld)
    seen = set()
    for k, batch in enumerate(ld):
        px = batch["pixel_values"]
        assert px.shape == (3, 3, 64, 64)
        for j in range(3):          # every output is the oracle transform of one of the stored images
            hit = [hw for hw, a in imgs.items()
                   if np.array_equal(px[j].numpy(), ipo.image_prep(a, 64, (ipo.smallest_max_size_dims(*hw, 64)[0] - 64) // 2,
                                                                   (ipo.smallest_max_size_dims(*hw, 64)[1] - 64) // 2, False))]
            assert len(hit) == 1
            seen.add(hit[0])
        if k == 11:
            break
    assert len(seen) >= 8           # resampled shards + shuffle buffer reach (almost) every image within 36 draws
