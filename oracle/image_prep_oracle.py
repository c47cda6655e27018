"""CPU restatement (numpy) of the reference's image pre-processing — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
path (e4t/data.py -> e4t_image_prep) never does.

What it restates:
  pretrain_e4t.py:137-144   make_transforms: albumentations.SmallestMaxSize(max_size, interpolation=3)
                            -> RandomCrop(size,size) -> HorizontalFlip(p=0.5)
  pretrain_e4t.py:174-177   image = (image / 127.5 - 1.0).astype(np.float32); HWC -> CHW
  (same transform in the iterable/webdataset branch, :292-300)

Third-party arithmetic that is NOT under /root/reference and is not installed here:
  * albumentations (requirements.txt: unpinned, 1.3.x in March 2023): SmallestMaxSize computes
    scale = max_size / min(h, w); if scale != 1: new dims = py3round(dim * scale) (round half to even);
    cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA).  RandomCrop: y1 = int((h - ch + 1) * u1),
    x1 = int((w - cw + 1) * u2) with u ~ U[0,1).
  * OpenCV cv2.resize(INTER_AREA) for 8-bit 3-channel images (imgproc/src/resize.cpp), restated below:
      - both scale factors (src/dst) >= 1:
          . both integer -> "area fast": 2x2 -> (a+b+c+d+2)>>2, else cvRound(int_sum * (1.f/area))
          . otherwise -> computeResizeAreaTab weights (double -> float), float accumulation first along x
            (buf += S*alpha, in table order) then along y (sum = beta*buf | sum += beta*buf), cvRound, saturate
      - otherwise (enlarging) -> the bilinear code path with the "area mode" coefficients, fixed point
        (11-bit coefficients, the 8-bit vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).
PARITY UNPINNED: neither cv2 nor albumentations can be imported in this environment and the reference
holds no golden images, so this restatement is checked only against properties (integer scales = exact box
means, constant images, PIL's BOX filter at integer scales) — see tests/test_data_host_logic.py.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def py3round(x: float) -> int:
    """albumentations.augmentations.geometric.functional.py3round: round half to even"""
    if abs(round(x) - x) == 0.5:
        return int(2.0 * round(x / 2.0))
    return int(round(x))


def smallest_max_size_dims(h: int, w: int, max_size: int):
    scale = max_size / float(min(w, h))
    if scale == 1.0:
        return h, w
    return py3round(h * scale), py3round(w * scale)


def random_crop_origin(h: int, w: int, size: int, u_h: float, u_w: float):
    """albumentations get_random_crop_coords"""
    return int((h - size + 1) * u_h), int((w - size + 1) * u_w)


# ----------------------------------------------------------------------------------------------------
def area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: per destination index a list of (source index, float32 weight), in table order"""
    tab = []
    for dx in range(dsize):
        ent = []
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def _padded(tab):
    n = max(len(e) for e in tab)
    si = np.zeros((len(tab), n), np.int64)
    al = np.zeros((len(tab), n), F32)
    for d, ent in enumerate(tab):
        for j, (s, a) in enumerate(ent):
            si[d, j], al[d, j] = s, a
    return si, al          # padding entries have weight 0: x + 0*S == x exactly, so the order of real terms is kept


def _resize_area_general(img, nh, nw):
    H, W, _ = img.shape
    sx, ax = _padded(area_tab(W, nw, W / nw))
    sy, ay = _padded(area_tab(H, nh, H / nh))
    S = img.astype(F32)
    buf = np.zeros((H, nw, 3), F32)
    for j in range(sx.shape[1]):                      # buf[dx] += S[sx]*alpha, table order
        buf = (buf + S[:, sx[:, j], :] * ax[None, :, j, None]).astype(F32)
    out = np.zeros((nh, nw, 3), F32)
    for j in range(sy.shape[1]):                      # sum = beta*buf ; sum += beta*buf
        out = (out + ay[:, j, None, None] * buf[sy[:, j]]).astype(F32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _resize_area_fast(img, nh, nw, fy, fx):
    H, W, _ = img.shape
    s = img.reshape(nh, fy, nw, fx, 3).astype(np.int64).sum(axis=(1, 3))
    if fx == 2 and fy == 2:
        return ((s + 2) >> 2).astype(np.uint8)
    v = s.astype(F32) * F32(F32(1.0) / F32(fx * fy))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def _linear_area_coeffs(ssize, dsize):
    scale = ssize / dsize
    inv = 1.0 / scale
    ofs = np.zeros(dsize, np.int64)
    co = np.zeros((dsize, 2), np.int64)
    for d in range(dsize):
        s = math.floor(d * scale)
        f = float(F32((d + 1) - (s + 1) * inv))
        f = 0.0 if f <= 0 else float(F32(f - math.floor(f)))
        if s < 0:
            f, s = 0.0, 0
        if s >= ssize - 1:
            f, s = 0.0, ssize - 1
        ofs[d] = s
        c0, c1 = F32(1.0) - F32(f), F32(f)
        co[d, 0] = int(np.clip(np.rint(F32(c0 * F32(2048))), -32768, 32767))
        co[d, 1] = int(np.clip(np.rint(F32(c1 * F32(2048))), -32768, 32767))
    return ofs, co


def _resize_linear_area(img, nh, nw):
    H, W, _ = img.shape
    xo, xa = _linear_area_coeffs(W, nw)
    yo, ya = _linear_area_coeffs(H, nh)
    S = img.astype(np.int64)
    x1 = np.minimum(xo + 1, W - 1)
    rows = S[:, xo, :] * xa[None, :, 0, None] + S[:, x1, :] * xa[None, :, 1, None]     # int, scaled by 2^11
    y1 = np.minimum(yo + 1, H - 1)
    r0, r1 = rows[yo], rows[y1]
    v = (((ya[:, 0, None, None] * (r0 >> 4)) >> 16) + ((ya[:, 1, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_inter_area(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """cv2.resize(img, (nw, nh), interpolation=cv2.INTER_AREA) for uint8 HxWx3"""
    H, W, _ = img.shape
    if (nh, nw) == (H, W):
        return img.copy()
    sx, sy = W / nw, H / nh
    if sx >= 1 and sy >= 1:
        ix, iy = int(np.rint(sx)), int(np.rint(sy))
        if abs(sx - ix) < np.finfo(np.float64).eps and abs(sy - iy) < np.finfo(np.float64).eps:
            return _resize_area_fast(img, nh, nw, iy, ix)
        return _resize_area_general(img, nh, nw)
    return _resize_linear_area(img, nh, nw)


def image_prep(img: np.ndarray, size: int, y0: int, x0: int, flip: bool) -> np.ndarray:
    """one sample of the reference's E4TDataset.__getitem__ with the random draws made explicit:
    uint8 HxWx3 -> float32 3 x size x size in [-1, 1]"""
    nh, nw = smallest_max_size_dims(img.shape[0], img.shape[1], size)
    r = resize_inter_area(img, nh, nw)
    c = r[y0:y0 + size, x0:x0 + size]
    if flip:
        c = c[:, ::-1]
    out = (c / 127.5 - 1.0).astype(np.float32)
    return np.ascontiguousarray(out.transpose(2, 0, 1))
