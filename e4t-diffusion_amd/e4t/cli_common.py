"""Host-side set-up shared by pretrain_e4t.py / tuning_e4t.py / inference.py: the glue lines of the reference scripts that sit
between "parse the arguments" and "run the step" — tokenizer + placeholder token, class-token / empty-prompt ids, prompt
templates, base-model loading — restated once so the three scripts condition the models identically (the E4T encoder pass is
conditioned on tokenizer("") and the class token in training AND in sampling; getting one of them different in one script
silently trains a model the pipeline cannot use).  No tensor arithmetic lives here.

There is no network in this build: "pretrained_model_name_or_path" is a LOCAL directory laid out as
    <dir>/unet.pt  vae.pt  text_encoder.pt        plain state dicts with the diffusers / transformers key names
    <dir>/unet_config.json                        optional (defaults to the --unet_variant architecture)
    <dir>/tokenizer/                              CLIPTokenizer files
    <dir>/scheduler/scheduler_config.json         optional
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Sequence

import torch

# the reference's prompt templates (pretrain_e4t.py:36-62 == tuning_e4t.py:28-54): data, kept verbatim so that a model trained
# here sees the prompt distribution a model trained with the reference sees
templates = [
    "a photo of {placeholder_token}",
    "the photo of {placeholder_token}",
    "a photo of a {placeholder_token}",
    "a photo of the {placeholder_token}",
    "a photo of one {placeholder_token}",
    "a close-up photo of the {placeholder_token}",
    "a bright photo of the {placeholder_token}",
    "a photo of a nice {placeholder_token}",
    "a good photo of {placeholder_token}",
    "a photo of a cool {placeholder_token}",
]
face_templates = templates + [
    "a portrait of {placeholder_token}",
    "the portrait of {placeholder_token}",
    "a portrait photo of {placeholder_token}",
    "portrait of {placeholder_token}",
    "portrait of the {placeholder_token}",
    "photo realistic portrait of {placeholder_token}",
]
art_templates = templates + [
    "art of {placeholder_token}",
    "art by {placeholder_token}",
]
NAMED_TEMPLATES = {"normal": templates, "face": face_templates, "art": art_templates}


def resolve_prompt_templates(prompt_template: str) -> List[str]:
    """pretrain_e4t.py:570-581 / tuning_e4t.py:254-265: a named list, or ONE custom template containing '{placeholder_token}'"""
    if prompt_template in NAMED_TEMPLATES:
        out = NAMED_TEMPLATES[prompt_template]
        print(f"Using the default {len(out)} templates!")
        return list(out)
    assert "{placeholder_token}" in prompt_template, "You must specify the location of placeholder token by '{placeholder_token}'"
    return [prompt_template]


def load_tokenizer(base_dir: Optional[str], allow_offline_standin: bool, vocab_size: int = 49408, max_len: int = 77):
    """CLIPTokenizer from <base_dir>/tokenizer (pretrain_e4t.py:233).  Without one — synthetic / random-init runs only — the
    offline whitespace tokenizer with CLIP's sizes stands in (its ids mean nothing to pretrained weights: loud message)."""
    tdir = os.path.join(base_dir or "", "tokenizer")
    if base_dir and os.path.isdir(tdir):
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(tdir)
    if not allow_offline_standin:
        raise FileNotFoundError(f"no tokenizer under {tdir!r}: real-data training needs the checkpoint's CLIPTokenizer files "
                                f"(pass --synthetic_data / --random_init for a run on random weights)")
    from .utils import WhitespaceTokenizer
    print("[e4t] no tokenizer directory: using the offline whitespace stand-in (only meaningful with randomly initialised weights)")
    return WhitespaceTokenizer(base_size=vocab_size, model_max_length=max_len)


def add_placeholder_token(tokenizer, text_encoder, placeholder_token: str) -> int:
    """pretrain_e4t.py:253-259 / tuning_e4t.py:121-127: add the token, grow the embedding table by it, return its id"""
    num_added = tokenizer.add_tokens(placeholder_token)
    if num_added == 0:
        raise ValueError(f"The tokenizer already contains the token {placeholder_token}. Please pass a different `placeholder_token` "
                         f"that is not already in the tokenizer.")
    pid = tokenizer.convert_tokens_to_ids(placeholder_token)
    text_encoder.resize_token_embeddings(len(tokenizer))
    assert text_encoder.get_input_embeddings().weight.shape[0] == len(tokenizer) and pid < len(tokenizer)
    return pid


def conditioning_ids(tokenizer, domain_class_token: str):
    """-> (class_token_id, empty_prompt_ids [1, max_len]).  pretrain_e4t.py:561-569: the class token must be ONE token; the E4T
    encoder pass is conditioned on tokenizer("") padded to the model length (BOS + EOS padding for CLIP)."""
    ids = tokenizer(domain_class_token, add_special_tokens=False, return_tensors="pt").input_ids[0]
    assert ids.shape[0] == 1, f"--domain_class_token {domain_class_token!r} must be a single token, got ids {ids.tolist()}"
    empty = tokenizer("", padding="max_length", truncation=True, max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    return int(ids[0]), empty


def tokenize_prompts(tokenizer, prompt_templates: Sequence[str], placeholder_token: str, placeholder_token_id: int, bsz: int, rng=random):
    """pretrain_e4t.py:607-615: one random template per sample -> (input_ids [bsz, max_len], index of the placeholder in each row)"""
    batch = rng.choices(list(prompt_templates), k=bsz)
    prompt = [t.format(placeholder_token=placeholder_token) for t in batch]
    ids = tokenizer(prompt, padding="max_length", truncation=True, max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    idx = torch.tensor([row.index(placeholder_token_id) for row in ids.tolist()])
    return ids, idx


def checked_load(module: torch.nn.Module, path: str, may_miss=lambda k: False, what: str = ""):
    """load_state_dict that fails loudly: unexpected keys always raise, missing keys raise unless `may_miss(key)` (the reference's
    load_e4t_unet / load_e4t_encoder contract, e4t/utils.py:119-124,150-154).  A shape mismatch raises inside torch."""
    sd = torch.load(path, map_location="cpu")
    # transformers < 4.31 saved the (persistent, integer) buffer `*.embeddings.position_ids` with every CLIP state dict; it is an
    # arange, not a weight, and the module trees here do not carry it: drop it, keep the strict check for every other key
    sd = {k: v for k, v in sd.items() if not k.endswith(".position_ids")}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not may_miss(k)]
    if missing or unexpected:
        raise RuntimeError(f"{what or path}: missing keys {missing[:5]}{'...' if len(missing) > 5 else ''} "
                           f"unexpected keys {list(unexpected)[:5]}{'...' if len(unexpected) > 5 else ''}")
    return module


def build_models(dev, base_dir: Optional[str], variant: str, seed: int, freeze_clip_vision: bool = True, e4t_dir: Optional[str] = None,
                 need_vae_encoder: bool = True):
    """(unet, e4t_encoder, text_encoder, vae_encoder) on `dev` (pretrain_e4t.py:233-251, tuning_e4t.py:97-118).
    base_dir: local Stable Diffusion checkpoint directory (see the module docstring) or None = random init from `seed`.
    e4t_dir: directory with weight_offsets.pt | unet.pt and encoder.pt of an earlier E4T run, loaded on top (both strict)."""
    from . import builders
    unet, enc, text, vae = builders.build_models(dev, variant, seed, freeze_clip_vision=freeze_clip_vision)
    if base_dir:
        if not os.path.isdir(base_dir):
            raise FileNotFoundError(f"{base_dir}: only local checkpoint directories are supported (no hub access in this build)")
        found = 0
        for name, mod, may_miss in (("unet", unet, lambda k: "wo" in k), ("vae", vae, lambda k: False), ("text_encoder", text, lambda k: False)):
            f = os.path.join(base_dir, f"{name}.pt")
            if os.path.exists(f):
                if name == "vae":       # an AutoencoderKL state dict also holds the decoder: keep the encoder half
                    sd = {k: v for k, v in torch.load(f, map_location="cpu").items() if k.startswith(("encoder.", "quant_conv."))}
                    missing, unexpected = mod.load_state_dict(sd, strict=False)
                    if missing or unexpected:
                        raise RuntimeError(f"{f}: missing {missing[:5]} unexpected {list(unexpected)[:5]}")
                else:
                    checked_load(mod, f, may_miss, what=f)
                found += 1
        if found == 0:
            raise FileNotFoundError(f"{base_dir} holds none of unet.pt / vae.pt / text_encoder.pt")
    if e4t_dir:
        for fn in ("weight_offsets.pt", "unet.pt"):
            f = os.path.join(e4t_dir, fn)
            if os.path.exists(f):
                checked_load(unet, f, (lambda k: "wo" not in k) if fn == "weight_offsets.pt" else (lambda k: False), what=f)
                print(f"Resuming from {f}")
        f = os.path.join(e4t_dir, "encoder.pt")
        if os.path.exists(f):
            checked_load(enc, f, what=f)
            print(f"Resuming from {f}")
    return unet, enc, text, vae
