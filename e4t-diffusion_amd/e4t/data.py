"""Training-image data path, MI355X-first.

Reference: ``pretrain_e4t.py:125-180`` (``_list_image_files_recursively``, ``make_transforms``, ``E4TDataset``) and the
``DataLoader`` at ``:284-291``.  There every sample is decoded, area-resized, cropped, flipped and normalised on ONE CPU
core per worker (``dataloader_num_workers`` defaults to 0) and the fp32 batch is then copied to the GPU.  At >100 images/s
per GPU that pipeline is the bottleneck, so here the split is different:

  host   : decode to uint8 RGB (PIL, a small thread pool — decoding releases the GIL), draw the random crop / flip,
           pack the RAW images of a batch into one pinned byte pool + an int64 plan table;
  device : one async H2D copy of pool + table on a copy stream, then ONE kernel (``e4t_image_prep``) does
           SmallestMaxSize(INTER_AREA) -> crop -> flip -> /127.5-1 -> CHW fp32 for the whole batch, computing only the
           cropped window.  Batches are prefetched (depth 2) so copy + prep hide under the previous training step.

Same public names and argument meaning as the reference (``make_transforms``, ``E4TDataset``); ``E4TDataset.__getitem__``
returns the raw image plus its transform plan instead of an already transformed tensor, and ``DeviceLoader`` replaces
``torch.utils.data.DataLoader`` + ``accelerator.prepare`` (rank r of ``world`` takes every world-th batch, as
accelerate's BatchSamplerShard does).  ``TarShardDataset`` is the ``--webdataset`` source (:303-340) without the webdataset
package: the loader pulls still-encoded images from resampled tar shards and decodes them in its thread pool.
"""
from __future__ import annotations

import os
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops

IMAGE_EXTENSIONS = ("jpg", "jpeg", "png", "gif")


def _list_image_files_recursively(data_dir):
    """sorted recursive listing of image files (pretrain_e4t.py:125-134; blobfile replaced by os)"""
    results = []
    for entry in sorted(os.listdir(data_dir)):
        full = os.path.join(data_dir, entry)
        ext = entry.split(".")[-1]
        if "." in entry and ext.lower() in IMAGE_EXTENSIONS:
            results.append(full)
        elif os.path.isdir(full):
            results.extend(_list_image_files_recursively(full))
    return results


def _py3round(x: float) -> int:
    if abs(round(x) - x) == 0.5:
        return int(2.0 * round(x / 2.0))
    return int(round(x))


class E4TTransform:
    """The reference's Compose([SmallestMaxSize(size, INTER_AREA), Center|RandomCrop(size, size), HorizontalFlip(0.5)])
    (pretrain_e4t.py:137-144) as a *plan*: the pixels are produced on the device by ``e4t_image_prep``."""

    def __init__(self, size: int, random_crop: bool = False, flip_p: float = 0.5):
        self.size, self.random_crop, self.flip_p = int(size), bool(random_crop), float(flip_p)

    def resized_dims(self, h: int, w: int):
        scale = self.size / float(min(h, w))
        if scale == 1.0:
            return h, w
        return _py3round(h * scale), _py3round(w * scale)

    def plan(self, h: int, w: int, rng=random):
        """(newH, newW, y0, x0, flip) for an image of h x w"""
        nh, nw = self.resized_dims(h, w)
        s = self.size
        if nh < s or nw < s:
            raise ValueError(f"image {h}x{w} resizes to {nh}x{nw}, smaller than the {s}x{s} crop")
        if self.random_crop:
            y0 = int((nh - s + 1) * rng.random())
            x0 = int((nw - s + 1) * rng.random())
        else:
            y0, x0 = (nh - s) // 2, (nw - s) // 2
        flip = int(rng.random() < self.flip_p)
        return nh, nw, y0, x0, flip


def make_transforms(size, random_crop=False):
    return E4TTransform(size, random_crop=random_crop)


class E4TDataset:
    """pretrain_e4t.py:147-180.  ``dataset_name``: a directory, several joined by "::", or a `datasets` name."""

    def __init__(self, dataset_name, resolution=512):
        from_datasets = False
        if os.path.isdir(dataset_name) or "::" in dataset_name:
            self.dataset = []
            for name in dataset_name.split("::"):
                self.dataset += _list_image_files_recursively(name)
        else:
            from datasets import load_dataset
            self.dataset = load_dataset(dataset_name, split="train")
            from_datasets = True
        self.from_datasets = from_datasets
        self.processor = make_transforms(resolution, random_crop=True)

    def __len__(self):
        return len(self.dataset)

    def load_rgb(self, idx) -> np.ndarray:
        image = self.dataset[idx]
        if self.from_datasets:
            image = image["image"]
        else:
            from PIL import Image
            image = Image.open(image)
        return np.ascontiguousarray(np.asarray(image.convert("RGB"), dtype=np.uint8))

    def __getitem__(self, idx, rng=random):
        image = self.load_rgb(idx)
        return dict(image=image, plan=self.processor.plan(image.shape[0], image.shape[1], rng))


def braceexpand(pattern: str):
    """The subset of bash brace expansion shard lists use (reference: braceexpand.braceexpand, pretrain_e4t.py:186):
    numeric ranges ``{000..127}`` (zero padding kept) and comma lists ``{a,b}``, nested / repeated left to right."""
    i = pattern.find("{")
    if i < 0:
        return [pattern]
    depth, j = 0, i
    while j < len(pattern):
        depth += pattern[j] == "{"
        depth -= pattern[j] == "}"
        if depth == 0:
            break
        j += 1
    if depth != 0:
        return [pattern]
    head, body, tail = pattern[:i], pattern[i + 1:j], pattern[j + 1:]
    if ".." in body and "," not in body and "{" not in body:
        lo, hi = body.split("..")[:2]
        width = max(len(lo), len(hi)) if (lo.startswith("0") or hi.startswith("0")) else 0
        step = 1 if int(hi) >= int(lo) else -1
        alts = [str(v).zfill(width) for v in range(int(lo), int(hi) + step, step)]
    else:
        alts, depth, cur = [], 0, ""
        for ch in body:
            if ch == "," and depth == 0:
                alts.append(cur)
                cur = ""
            else:
                depth += ch == "{"
                depth -= ch == "}"
                cur += ch
        alts.append(cur)
        if len(alts) == 1:                          # "{x}" is literal in bash
            return [head + "{" + a + "}" + t for a in braceexpand(body) for t in braceexpand(tail)]
    return [head + a2 + t for a in alts for a2 in braceexpand(a) for t in braceexpand(tail)]


def get_dataset_size(shards: str):
    """(number of samples or None, number of shards) from sizes.json / <shard>_stats.json (pretrain_e4t.py:183-211)"""
    import json
    shards_list = []
    for s in shards.split("::"):
        shards_list += braceexpand(s)
    sizes_filename = os.path.join(os.path.dirname(shards), "sizes.json")
    if os.path.exists(sizes_filename):
        with open(sizes_filename) as f:
            sizes = json.load(f)
        total = sum(int(sizes[os.path.basename(shard)]) for shard in shards_list)
    else:
        total = 0
        for shard in shards_list:
            jp = shard.replace(".tar", "_stats.json")
            if os.path.exists(jp):
                with open(jp) as f:
                    st = json.load(f)
                total += int(st["n_data"] if "n_data" in st else st["successes"])
            else:
                print(f"Not Found {jp}")
    return total, len(shards_list)


class TarShardDataset:
    """The reference's ``--webdataset`` source (pretrain_e4t.py:303-318) on the standard library: tar shards whose members
    ``<key>.<ext>`` form samples; shards are drawn at random with replacement for ever (``wds.ResampledShards``), samples
    without a ``jpg`` member are skipped (``filter_webdataset``), a shuffle buffer of 1000 decorrelates neighbours
    (``wds.shuffle(1000)``), unreadable shards / members are reported and skipped (``wds.warn_and_continue``).  Items are
    the still-encoded image bytes: decoding happens in the loader's thread pool, resize/crop/flip/normalise on the GPU."""

    def __init__(self, shards: str, resolution=512, shuffle_buffer=1000, image_key="jpg"):
        self.shards = []
        for s in shards.split("::"):
            self.shards += braceexpand(s)
        if not self.shards:
            raise ValueError("no shards")
        self.processor = make_transforms(resolution, random_crop=True)
        self.shuffle_buffer, self.image_key = shuffle_buffer, image_key

    def _samples(self, path):
        import tarfile
        try:
            with tarfile.open(path) as tf:
                key, cur = None, {}
                for m in tf:
                    if not m.isfile():
                        continue
                    base = os.path.basename(m.name)
                    k, _, ext = base.partition(".")
                    k = os.path.join(os.path.dirname(m.name), k)
                    if k != key:
                        if cur:
                            yield cur
                        key, cur = k, {}
                    cur[ext.lower()] = tf.extractfile(m).read()
                if cur:
                    yield cur
        except Exception as e:                      # warn_and_continue
            print(f"[TarShardDataset] skipping {path}: {e!r}")

    def iter_items(self, rank=0, world=1, seed=0, epoch=0):
        rng = random.Random((seed * 1000003 + epoch) * 4099 + rank)
        buf = []
        while True:
            for smp in self._samples(rng.choice(self.shards)):
                data = smp.get(self.image_key)
                if data is None:
                    continue
                if len(buf) < self.shuffle_buffer:
                    buf.append(data)
                    continue
                j = rng.randrange(len(buf))
                buf[j], data = data, buf[j]
                yield data
            if buf and len(buf) < self.shuffle_buffer and len(self.shards) == 1:
                rng.shuffle(buf)                    # tiny single-shard sets: do not starve, hand the buffer out
                yield from buf
                buf = []

    def __getitem__(self, item, rng=random):
        import io

        from PIL import Image
        image = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(item)).convert("RGB"), dtype=np.uint8))
        return dict(image=image, plan=self.processor.plan(image.shape[0], image.shape[1], rng))


class _Slot:
    """one in-flight batch: pinned staging + device buffers (grown on demand, reused)"""

    def __init__(self, device):
        self.device = device
        self.h_pool = self.d_pool = None
        self.h_table = self.d_table = None
        self.out = None
        self.event = None

    def ensure(self, nbytes, B, S):
        pin = self.device.type == "cuda"
        if self.h_pool is None or self.h_pool.numel() < nbytes:
            cap = int(nbytes * 1.25) + 4096
            self.h_pool = torch.empty(cap, dtype=torch.uint8, pin_memory=pin)
            self.d_pool = torch.empty(cap, dtype=torch.uint8, device=self.device)
        if self.h_table is None or self.h_table.shape[0] != B:
            self.h_table = torch.empty((B, 8), dtype=torch.int64, pin_memory=pin)
            self.d_table = torch.empty((B, 8), dtype=torch.int64, device=self.device)


def pack_batch(samples, size, h_pool=None, h_table=None):
    """samples: [{image: uint8 HxWx3, plan: (nh, nw, y0, x0, flip)}] -> (pool bytes, int64 [B,8] table) (host tensors).
    Image offsets are 16-byte aligned."""
    offs, total = [], 0
    for s in samples:
        offs.append(total)
        total += (s["image"].size + 15) // 16 * 16
    if h_pool is None:
        h_pool = torch.empty(total, dtype=torch.uint8)
    if h_table is None:
        h_table = torch.empty((len(samples), 8), dtype=torch.int64)
    pool_np = h_pool.numpy()
    for i, (s, off) in enumerate(zip(samples, offs)):
        img = s["image"]
        H, W, C = img.shape
        assert C == 3 and img.dtype == np.uint8
        nh, nw, y0, x0, flip = s["plan"]
        if not (0 <= y0 and y0 + size <= nh and 0 <= x0 and x0 + size <= nw):
            raise ValueError(f"crop window ({y0},{x0})+{size} outside the resized image {nh}x{nw}")
        pool_np[off:off + img.size] = img.reshape(-1)
        h_table[i] = torch.tensor([off, H, W, nh, nw, y0, x0, flip], dtype=torch.int64)
    return h_pool, h_table, total


class DeviceLoader:
    """Iterates ``dict(pixel_values=fp32 [B,3,S,S] on `device`)`` over an ``E4TDataset``.

    shuffle=True reshuffles every epoch with `seed` (all ranks draw the same permutation; rank r takes batches
    r, r+world, ...; the incomplete last batch is dropped).  num_workers = decode threads (0 -> decode inline in the
    prefetch thread).  The tensors handed out are valid on the consumer's current stream; they are reused `prefetch`
    batches later, so consume (or clone) them within the step, as a training step does."""

    def __init__(self, dataset, batch_size, shuffle=True, num_workers=0, device="cuda", rank=0, world=1, seed=0,
                 prefetch=2, drop_last=True):
        self.ds, self.B = dataset, int(batch_size)
        self.shuffle, self.rank, self.world, self.seed = shuffle, rank, world, seed
        self.device = torch.device(device)
        self.size = dataset.processor.size
        self.prefetch = max(1, int(prefetch))
        self.pool = ThreadPoolExecutor(num_workers) if num_workers > 0 else None
        # one batch with the consumer + `prefetch` queued + one being produced
        self.slots = [_Slot(self.device) for _ in range(self.prefetch + 2)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.epoch = 0
        if not drop_last:
            raise NotImplementedError("a ragged last batch is not supported (the reference's webdataset loader also drops it)")

    def __len__(self):
        if hasattr(self.ds, "iter_items"):
            raise TypeError("an iterable (tar-shard) source has no length: it is resampled for ever")
        return len(self.ds) // (self.B * self.world)

    def _indices(self):
        if hasattr(self.ds, "iter_items"):          # stream of still-encoded items, B at a time; bad images are skipped in _produce
            it = self._items = self.ds.iter_items(self.rank, self.world, self.seed, self.epoch)
            while True:
                yield [next(it) for _ in range(self.B)]
        n = len(self.ds)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
            perm = torch.randperm(n, generator=g).tolist()
        else:
            perm = list(range(n))
        nb = n // (self.B * self.world)
        for i in range(nb):
            b = i * self.world + self.rank
            yield perm[b * self.B:(b + 1) * self.B]

    def _load(self, item, rng):
        try:
            return self.ds.__getitem__(item, rng)
        except Exception as e:
            if not hasattr(self.ds, "iter_items"):
                raise
            print(f"[DeviceLoader] skipping a sample: {e!r}")          # wds.warn_and_continue
            return None

    def _produce(self, slot, idxs, rng):
        plans_rng = [random.Random(rng.getrandbits(64)) for _ in idxs]        # per-sample streams: thread-order independent
        if self.pool is not None:
            samples = list(self.pool.map(lambda a: self._load(a[0], a[1]), zip(idxs, plans_rng)))
        else:
            samples = [self._load(i, r) for i, r in zip(idxs, plans_rng)]
        for k in range(len(samples)):               # stream sources: an undecodable image is replaced by the next item
            while samples[k] is None:
                samples[k] = self._load(next(self._items), plans_rng[k])
        nbytes = sum((s["image"].size + 15) // 16 * 16 for s in samples)
        with ops.capture_lock:       # (host decode above ran unlocked) no allocation / copy / event wait while a HIP graph is being captured
            return self._upload(slot, samples, nbytes)

    def _upload(self, slot, samples, nbytes):
        if slot.event is not None:
            slot.event.synchronize()                     # the consumer's step that used this slot's output has been enqueued and finished
        slot.ensure(nbytes, self.B, self.size)
        _, _, total = pack_batch(samples, self.size, slot.h_pool, slot.h_table)
        be = ops.backend()
        if self.copy_stream is None:
            slot.d_pool[:total].copy_(slot.h_pool[:total])
            slot.d_table.copy_(slot.h_table)
            slot.out = be.image_prep(slot.d_pool, slot.d_table, self.B, self.size, out=slot.out)
            return None
        with torch.cuda.stream(self.copy_stream):
            slot.d_pool[:total].copy_(slot.h_pool[:total], non_blocking=True)
            slot.d_table.copy_(slot.h_table, non_blocking=True)
            slot.out = be.image_prep(slot.d_pool, slot.d_table, self.B, self.size, out=slot.out)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return ready

    def __iter__(self):
        rng = random.Random(self.seed * 7919 + self.epoch * 104729 + self.rank)
        batches = self._indices() if hasattr(self.ds, "iter_items") else list(self._indices())
        self.epoch += 1
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        dev = self.device

        def worker():
            try:
                if dev.type == "cuda":
                    torch.cuda.set_device(dev)
                for k, idxs in enumerate(batches):
                    if stop.is_set():
                        return
                    slot = self.slots[k % len(self.slots)]
                    ready = self._produce(slot, idxs, rng)
                    q.put((slot, ready))
                q.put(None)
            except BaseException as e:          # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                slot, ready = item
                if ready is not None:
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(ready)
                yield dict(pixel_values=slot.out)
                if ready is not None:
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(dev))   # the slot may be rewritten once the consumer's work so far is done
                    slot.event = done
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
