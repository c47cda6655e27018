"""Module TREES (parameter containers, no arithmetic) of the components around the UNet: the CLIP text encoder (SURVEY.md §2
#8), the AutoencoderKL encoder (#15) and decoder.  They fix the parameter names / shapes of the HF / diffusers checkpoints
(CLIPTextModel, AutoencoderKL.encoder / .decoder / quant convs) so that real weights load by key; the kernel-driven classes
in ``text.py`` and ``vae.py`` derive from them and add the forward passes.  Nothing here computes: the stock-torch
forward passes of the same trees, used as fp32 parity references, live with the tests (tests/torch_twins.py).

CLIPTextModel accepts ``inputs_embeds`` like the reference's patched class (e4t/models/modeling_clip.py:9-82) — the
installed transformers 5.x no longer has the internals that file monkey-patches, so an equivalent is needed anyway.
"""
from __future__ import annotations

import torch
from torch import nn

CLIP_TEXT_L = dict(vocab_size=49409, hidden_size=768, num_layers=12, num_heads=12, intermediate_size=3072, max_len=77, act="quick_gelu")
CLIP_TEXT_H = dict(vocab_size=49409, hidden_size=1024, num_layers=23, num_heads=16, intermediate_size=4096, max_len=77, act="gelu")


class _SelfAttn(nn.Module):
    def __init__(self, w, heads):
        super().__init__()
        self.heads = heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(w, w) for _ in range(4))


class _Mlp(nn.Module):
    def __init__(self, w, inter, act):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(w, inter), nn.Linear(inter, w), act


class _Layer(nn.Module):
    def __init__(self, w, heads, inter, act):
        super().__init__()
        self.self_attn = _SelfAttn(w, heads)
        self.layer_norm1 = nn.LayerNorm(w)
        self.mlp = _Mlp(w, inter, act)
        self.layer_norm2 = nn.LayerNorm(w)


class _Embeddings(nn.Module):
    def __init__(self, vocab, w, max_len):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, w)
        self.position_embedding = nn.Embedding(max_len, w)


class _Encoder(nn.Module):
    def __init__(self, w, layers, heads, inter, act):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(w, heads, inter, act) for _ in range(layers)])


class _TextTransformer(nn.Module):
    def __init__(self, vocab_size, hidden_size, num_layers, num_heads, intermediate_size, max_len, act):
        super().__init__()
        self.embeddings = _Embeddings(vocab_size, hidden_size, max_len)
        self.encoder = _Encoder(hidden_size, num_layers, num_heads, intermediate_size, act)
        self.final_layer_norm = nn.LayerNorm(hidden_size)


class CLIPTextModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        cfg = dict(CLIP_TEXT_L, **cfg)
        self.config = cfg
        self.text_model = _TextTransformer(**cfg)

    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    @property
    def device(self):            # transformers' PreTrainedModel surface the reference's pipeline reads (pipeline_stable_diffusion_e4t.py:60,83)
        return self.text_model.embeddings.token_embedding.weight.device

    @property
    def dtype(self):
        return self.text_model.embeddings.token_embedding.weight.dtype

    def resize_token_embeddings(self, new_num_tokens: int):
        """transformers' PreTrainedModel.resize_token_embeddings for the input embedding: keep the existing rows, new rows
        N(0, 0.02) (the reference grows the table by the placeholder token, pipeline_stable_diffusion_e4t.py:53)"""
        old = self.text_model.embeddings.token_embedding
        n_old, w = old.weight.shape
        if new_num_tokens == n_old:
            return old
        new = nn.Embedding(new_num_tokens, w, device=old.weight.device, dtype=old.weight.dtype)
        new.weight.data.normal_(mean=0.0, std=0.02)
        k = min(n_old, new_num_tokens)
        new.weight.data[:k] = old.weight.data[:k]
        new.weight.requires_grad_(old.weight.requires_grad)
        self.text_model.embeddings.token_embedding = new
        self.config["vocab_size"] = new_num_tokens
        return new

    def forward(self, input_ids=None, inputs_embeds=None):
        raise NotImplementedError("parameter tree only: use e4t.text.CLIPTextModel (HIP kernels)")


# ------------------------------------------------------------------------------------------------
class _VRes(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _VDown(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)


class _VDownBlock(nn.Module):
    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([_VRes(cin, cout), _VRes(cout, cout)])
        self.downsamplers = nn.ModuleList([_VDown(cout)]) if down else None


class _VAttn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))


class _VMid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([_VRes(c, c), _VRes(c, c)])
        self.attentions = nn.ModuleList([_VAttn(c)])


class _VEncoder(nn.Module):
    def __init__(self, boc, latent):
        super().__init__()
        self.conv_in = nn.Conv2d(3, boc[0], 3, padding=1)
        blocks, c = [], boc[0]
        for i, co in enumerate(boc):
            blocks.append(_VDownBlock(c, co, i < len(boc) - 1))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _VMid(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)


class VAEEncoder(nn.Module):
    """Parameter tree of AutoencoderKL.encode(x).latent_dist.sample() * scaling_factor  (pretrain_e4t.py:598-599)."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, scaling_factor=0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.encoder = _VEncoder(tuple(block_out_channels), latent_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)


# ------------------------------------------------------------------------------------------------
class _VUp(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class _VUpBlock(nn.Module):
    def __init__(self, cin, cout, n, up):
        super().__init__()
        self.resnets = nn.ModuleList([_VRes(cin if j == 0 else cout, cout) for j in range(n)])
        self.upsamplers = nn.ModuleList([_VUp(cout)]) if up else None


class _VDecoder(nn.Module):
    def __init__(self, boc, latent, out_channels, layers_per_block):
        super().__init__()
        rev = tuple(reversed(boc))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _VMid(rev[0])
        blocks, c = [], rev[0]
        for i, co in enumerate(rev):
            blocks.append(_VUpBlock(c, co, layers_per_block + 1, i < len(rev) - 1))
            c = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)


class VAEDecoder(nn.Module):
    """Parameter tree of AutoencoderKL.decode(latents / scaling_factor).sample and the pipeline's decode_latents
    (pipeline_stable_diffusion_e4t.py:226,237); key names of the diffusers checkpoint (post_quant_conv, decoder.*)."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, out_channels=3, layers_per_block=2,
                 scaling_factor=0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.block_out_channels = tuple(block_out_channels)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _VDecoder(tuple(block_out_channels), latent_channels, out_channels, layers_per_block)
