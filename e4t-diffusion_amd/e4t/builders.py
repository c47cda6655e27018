"""Model construction for the scripts and the benchmark: the architectures BASELINE.json names, built on the native
modules (random init when no checkpoint directory is given — there is no network here for the hub weights the reference
downloads at pretrain_e4t.py:233-251)."""
from __future__ import annotations

import torch

SD_UNET_BASE = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                    norm_num_groups=32, norm_eps=1e-5)
# CompVis/stable-diffusion-v1-4 (unet/config.json) and stabilityai/stable-diffusion-2-1 at 768 px
UNET_CONFIGS = {
    "sd14": dict(SD_UNET_BASE, cross_attention_dim=768, attention_head_dim=8),
    "sd21": dict(SD_UNET_BASE, sample_size=96, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20), use_linear_projection=True,
                 upcast_attention=True),
}
# text_encoder/config.json of the same checkpoints (CLIP ViT-L/14 text tower; OpenCLIP ViT-H text tower, 23 layers as shipped)
CLIP_TEXT_L = dict(vocab_size=49408, hidden_size=768, num_layers=12, num_heads=12, intermediate_size=3072, max_len=77, act="quick_gelu")
CLIP_TEXT_H = dict(vocab_size=49408, hidden_size=1024, num_layers=23, num_heads=16, intermediate_size=4096, max_len=77, act="gelu")
TEXT_CONFIGS = {"sd14": CLIP_TEXT_L, "sd21": CLIP_TEXT_H}
RESOLUTION = {"sd14": 512, "sd21": 768}
# a 4-level toy of the same topology: the scripts' set-up code is exercised with it on machines without a GPU (tests/test_cli_setup.py)
UNET_CONFIGS["tiny-test"] = dict(SD_UNET_BASE, sample_size=16, block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=2)
TEXT_CONFIGS["tiny-test"] = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=128, max_len=9, act="quick_gelu")
RESOLUTION["tiny-test"] = 64


def build_models(dev, model="sd14", seed=0, vocab_size=None, freeze_clip_vision=True):
    """(unet, e4t_encoder, text_encoder, vae_encoder) of `model` on `dev`, random init from `seed`.  `vocab_size`: rows of the
    token-embedding table (the checkpoint's 49408 + the placeholder token the scripts add, pretrain_e4t.py:254-259)."""
    from .encoder import E4TEncoder
    from .models.unet_2d_condition import UNet2DConditionModel
    from .text import CLIPTextModel
    from .vae import VAEEncoder
    torch.manual_seed(seed)
    ucfg, tcfg = UNET_CONFIGS[model], dict(TEXT_CONFIGS[model])
    if vocab_size is not None:
        tcfg["vocab_size"] = vocab_size
    with torch.device(dev):
        unet = UNet2DConditionModel(**ucfg)
        tiny = model == "tiny-test"
        enc = E4TEncoder(word_embedding_dim=tcfg["hidden_size"], block_out_channels=ucfg["block_out_channels"],
                         arch="ViT-tiny-test" if tiny else "ViT-H-14", freeze_clip_vision=freeze_clip_vision, **(dict(n_odd_layers=3) if tiny else {}))
        text = CLIPTextModel(**tcfg).requires_grad_(False)      # fp32 master weights; bf16 compute copies are made once
        vae = VAEEncoder(**(dict(block_out_channels=(64, 64)) if tiny else {})).requires_grad_(False)
    return unet, enc, text, vae
