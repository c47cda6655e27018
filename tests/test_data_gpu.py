"""-m gpu: the device data path end to end — PNG files -> DeviceLoader (pinned pool, copy stream, e4t_image_prep kernel,
prefetch) -> byte-exact against the numpy oracle."""
import numpy as np
import pytest
import torch
from PIL import Image

import image_prep_oracle as ipo

pytestmark = pytest.mark.gpu


def test_device_loader_matches_oracle(hip_env, tmp_path):
    from e4t.data import DeviceLoader, E4TDataset
    hip, emu, dev, ops = hip_env
    rng = np.random.default_rng(11)
    dims = [(300, 420), (512, 512), (1024, 768), (700, 933), (256, 300), (640, 512), (513, 1000), (1536, 1536), (999, 777), (520, 530)]
    root = tmp_path / "imgs"
    root.mkdir()
    imgs = {}
    for i, (h, w) in enumerate(dims):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        f = root / f"{i:02d}.png"
        Image.fromarray(a).save(f)
        imgs[str(f)] = a
    ds = E4TDataset(str(root), resolution=256)
    plans = {}
    orig = ds.processor.plan

    def recording_plan(h, w, rng=None):
        p = orig(h, w, rng)
        plans.setdefault((h, w), []).append(p)
        return p

    ds.processor.plan = recording_plan
    ld = DeviceLoader(ds, batch_size=3, shuffle=True, num_workers=3, device=dev, seed=1)
    for epoch in range(2):
        order = [i for b in ld._indices() for i in b]
        k = 0
        for batch in ld:
            px = batch["pixel_values"]
            assert px.is_cuda and px.shape == (3, 3, 256, 256)
            got = px.cpu().numpy()
            for j in range(3):
                a = imgs[ds.dataset[order[k]]]
                nh, nw, y0, x0, flip = plans[a.shape[:2]][epoch]
                np.testing.assert_array_equal(got[j], ipo.image_prep(a, 256, y0, x0, bool(flip)))
                k += 1
        assert k == 9
