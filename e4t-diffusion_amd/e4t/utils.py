"""Small host-side helpers that mirror the reference's e4t/utils.py surface (config dict access,
checkpoint key filters).  No tensor arithmetic lives here."""
from __future__ import annotations

import json
import os

import torch


class AttributeDict(dict):
    """dict with attribute access; missing keys read as None (reference: e4t/utils.py:17-40)."""

    def __getattr__(self, k):
        return self.get(k, None)

    def __setattr__(self, k, v):
        self[k] = v


def weight_offset_state_dict(unet) -> dict:
    """The ``weight_offsets.pt`` payload: every UNet state-dict entry whose key contains "wo" (e4t/utils.py:129-131)."""
    return {k: v for k, v in unet.state_dict().items() if "wo" in k}


def save_e4t_unet(unet, save_dir):
    os.makedirs(save_dir, exist_ok=True)
    torch.save(weight_offset_state_dict(unet), os.path.join(save_dir, "weight_offsets.pt"))


def load_weight_offsets(unet, path):
    sd = torch.load(path, map_location="cpu")
    missing, unexpected = unet.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "wo" in k]
    if bad or unexpected:
        raise RuntimeError(f"weight_offsets.pt does not match the UNet: missing {bad[:3]} unexpected {unexpected[:3]}")


def save_e4t_encoder(encoder, save_dir):
    os.makedirs(save_dir, exist_ok=True)
    torch.save(encoder.state_dict(), os.path.join(save_dir, "encoder.pt"))


def save_config(args_dict, save_dir):
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, "config.json"), "w") as f:
        json.dump(args_dict, f, indent=2, default=str)


def load_config_from_pretrained(pretrained_model_name_or_path) -> AttributeDict:
    """config.json of an E4T checkpoint directory (reference e4t/utils.py:75-89).  Hub model names (the reference's MODELS
    table) need network access, which this build never assumes: only local paths are accepted."""
    path = pretrained_model_name_or_path
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: only local checkpoint directories are supported (no hub download in this build)")
    if "config.json" not in path:
        path = os.path.join(path, "config.json")
    with open(path, "r", encoding="utf-8") as f:
        return AttributeDict(json.load(f))


def load_e4t_unet(pretrained_model_name_or_path=None, ckpt_path=None, unet_config=None, **kwargs):
    """Reference e4t/utils.py:92-126: base UNet weights + (optionally) ``weight_offsets.pt`` / ``unet.pt`` on top.
    The base weights come from ``<pretrained_model_name_or_path>/unet.pt`` (a plain state dict; there is no diffusers
    ``from_pretrained`` here) and the architecture from ``unet_config`` or ``<...>/unet_config.json``.  Raises on missing
    keys when a checkpoint is given and on unexpected keys always, like the reference."""
    from .models.unet_2d_condition import UNet2DConditionModel
    assert pretrained_model_name_or_path is not None or ckpt_path is not None
    if ckpt_path is not None:
        assert os.path.basename(ckpt_path) in ("unet.pt", "weight_offsets.pt"), "You must specify the filename! (`unet.pt` or `weight_offsets.pt`)"
        if pretrained_model_name_or_path is None:
            config = load_config_from_pretrained(os.path.dirname(ckpt_path))
            pretrained_model_name_or_path = (config.pretrained_args or {}).get("pretrained_model_name_or_path") or config.pretrained_model_name_or_path
    if unet_config is None:
        with open(os.path.join(pretrained_model_name_or_path, "unet_config.json")) as f:
            unet_config = json.load(f)
    unet = UNet2DConditionModel(**{k: v for k, v in unet_config.items() if not k.startswith("_")}, **kwargs)
    state = dict(unet.state_dict())
    base = os.path.join(pretrained_model_name_or_path or "", "unet.pt")
    if os.path.exists(base):
        state.update(torch.load(base, map_location="cpu"))
    if ckpt_path:
        state.update(torch.load(ckpt_path, map_location="cpu"))
        print(f"Resuming from {ckpt_path}")
    m, u = unet.load_state_dict(state, strict=False)
    if len(m) > 0 and ckpt_path:
        raise RuntimeError(f"missing keys:\n{m}")
    if len(u) > 0:
        raise RuntimeError(f"unexpected keys:\n{u}")
    return unet


def load_e4t_encoder(ckpt_path=None, **kwargs):
    """Reference e4t/utils.py:134-155 (local paths only)."""
    from .encoder import E4TEncoder
    encoder = E4TEncoder(**kwargs)
    if ckpt_path:
        if not os.path.exists(ckpt_path):
            raise FileNotFoundError(f"{ckpt_path}: only local checkpoints are supported (no hub download in this build)")
        if "encoder.pt" not in ckpt_path:
            ckpt_path = os.path.join(ckpt_path, "encoder.pt")
        state = torch.load(ckpt_path, map_location="cpu")
        print(f"Resuming from {ckpt_path}")
        m, u = encoder.load_state_dict(state, strict=False)
        if len(m) > 0:
            raise RuntimeError(f"missing keys:\n{m}")
        if len(u) > 0:
            raise RuntimeError(f"unexpected keys:\n{u}")
    return encoder


def load_image(image_path, resolution=None):
    """Reference e4t/utils.py:171-178: RGB PIL image; with ``resolution`` the SmallestMaxSize + centre-cropped version.
    Host-side helper for the demo / CLI (one image): PIL's BOX filter stands in for cv2.INTER_AREA here — identical box
    means at integer factors, within rounding otherwise; the training data path uses the byte-exact kernel instead."""
    from PIL import Image
    img = Image.open(image_path).convert("RGB")
    if resolution:
        w, h = img.size
        scale = resolution / min(w, h)
        nw, nh = max(resolution, round(w * scale)), max(resolution, round(h * scale))
        img = img.resize((nw, nh), Image.BOX if scale < 1 else Image.BILINEAR)
        x0, y0 = (nw - resolution) // 2, (nh - resolution) // 2
        img = img.crop((x0, y0, x0 + resolution, y0 + resolution))
    return img


def image_grid(imgs, rows, cols):
    """Reference e4t/utils.py:181-191"""
    from PIL import Image
    assert len(imgs) == rows * cols
    w, h = imgs[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, img in enumerate(imgs):
        grid.paste(img, box=(i % cols * w, i // cols * h))
    return grid


class WhitespaceTokenizer:
    """Offline stand-in with the CLIPTokenizer call surface the pipeline uses (no vocabulary files can be fetched here):
    whitespace words -> ids (unknown words are added on the fly), BOS/EOS, padding to model_max_length.  For smoke runs,
    tests and benchmarks with random-init weights only."""
    model_max_length = 9

    def __init__(self, base_size: int = 100, model_max_length: int = 9):
        self.base_size, self.model_max_length = base_size, model_max_length      # CLIP: 49408 tokens, 77 positions
        self.vocab = {"<bos>": 1, "<eos>": 2, "a": 5, "photo": 6, "of": 7, "art": 11, "painting": 12}

    def __len__(self):
        return self.base_size + sum(1 for v in self.vocab.values() if v >= self.base_size)

    def add_tokens(self, tok):
        if tok in self.vocab:
            return 0
        self.vocab[tok] = len(self)
        return 1

    def convert_tokens_to_ids(self, tok):
        return self.vocab[tok]

    def __call__(self, text, padding=None, truncation=None, max_length=None, return_tensors=None, add_special_tokens=True):
        texts = [text] if isinstance(text, str) else text
        rows = []
        for t in texts:
            ids = [self.vocab[w] if w in self.vocab else self.vocab.setdefault(w, 20 + (sum(map(ord, w)) % 70)) for w in t.split()]
            if add_special_tokens:
                ids = [1] + ids + [2]
            if padding == "max_length":
                ids = (ids + [2] * max_length)[:max_length]
            rows.append(ids)

        class R:
            input_ids = torch.tensor(rows, dtype=torch.long)
        return R()
