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
