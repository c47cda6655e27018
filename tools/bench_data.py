"""Data-path numbers on the GPU box: (1) the e4t_image_prep kernel alone at B=16, 512 px over WikiArt-like image sizes,
(2) DeviceLoader images/s from JPEG files with N decode threads, (3) the numpy oracle of the same transform on one core."""
import io, os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "e4t-diffusion_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from PIL import Image
from e4t import ops
from e4t.data import DeviceLoader, E4TDataset, pack_batch, make_transforms
import image_prep_oracle as ipo

dev = torch.device("cuda:0")
be = ops.backend()
rng = np.random.default_rng(0)
dims = [(1382, 1024), (1024, 1280), (2000, 1600), (768, 1024), (1500, 1500), (1024, 1024), (900, 1400), (3000, 2400)] * 2
tf = make_transforms(512, random_crop=True)
import random
samples = []
for h, w in dims:
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    samples.append(dict(image=img, plan=tf.plan(h, w, random.Random(1))))
pool, table, total = pack_batch(samples, 512)
dp, dt = pool.to(dev), table.to(dev)
out = be.image_prep(dp, dt, 16, 512)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    be.image_prep(dp, dt, 16, 512, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
win = sum(512 * 512 * (h / p[0]) * (w / p[1]) * 3 for (h, w), p in zip(dims, [s["plan"] for s in samples]))
print(f"image_prep B=16 512px: {ms*1e3:.1f} us/batch, source bytes {total/1e6:.1f} MB (window {win/1e6:.1f} MB) + {16*3*512*512*4/1e6:.1f} MB out "
      f"-> {(win + 16*3*512*512*4)/ms/1e6:.1f} GB/s algorithmic, {16/ms*1e3:.0f} img/s")
t0 = time.perf_counter()
for s in samples[:4]:
    ipo.image_prep(s["image"], 512, s["plan"][2], s["plan"][3], bool(s["plan"][4]))
print(f"numpy oracle: {(time.perf_counter()-t0)/4*1e3:.1f} ms/image on one core")
with tempfile.TemporaryDirectory() as d:
    for i, s in enumerate(samples * 4):
        sm = np.asarray(Image.fromarray(s["image"]).resize((s["image"].shape[1] // 8, s["image"].shape[0] // 8)).resize((s["image"].shape[1], s["image"].shape[0]), Image.BICUBIC))
        Image.fromarray(sm).save(os.path.join(d, f"{i:03d}.jpg"), quality=90)
    ds = E4TDataset(d, resolution=512)
    for nw in (0, 4, 8, 16):
        ld = DeviceLoader(ds, 16, num_workers=nw, device=dev)
        list(ld)
        t0 = time.perf_counter(); n = 0
        for _ in range(2):
            for b in ld:
                n += b["pixel_values"].shape[0]
        torch.cuda.synchronize()
        print(f"DeviceLoader num_workers={nw}: {n/(time.perf_counter()-t0):.0f} img/s (JPEG decode on the host, {os.cpu_count()} cores)")
