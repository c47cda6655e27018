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
