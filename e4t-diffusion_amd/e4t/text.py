"""CLIP text encoder with ``inputs_embeds`` on the HIP kernels — SURVEY.md §8f row N3 (reference:
e4t/models/modeling_clip.py:9-82 patches transformers' CLIPTextModel so the E4T domain embedding can be written into the
token embeddings; pretrain_e4t.py:616,630-634 run it between the two UNet passes, forward + gradient w.r.t. the embeddings).

Same module tree / parameter names as the stock-torch twin in ``frozen.py`` (= the HF checkpoint keys), so weights load by
key into either.  The weights are frozen in both reference scripts; the native path therefore builds a fused q|k|v weight
per layer once and propagates only dX: LayerNorm (residual gradient folded in) -> one (tokens x 3w) GEMM -> causal fused
attention -> out-proj GEMM with the residual in its epilogue -> LayerNorm -> fc1 -> quick_gelu / gelu -> fc2 (+ residual).
A trainable text encoder falls back to the torch twin (never used by the reference)."""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .frozen import CLIPTextModel as _TorchCLIPTextModel


class CLIPTextModel(_TorchCLIPTextModel):
    def __init__(self, **cfg):
        super().__init__(**cfg)
        self._fused = None

    def _prepare(self):
        """Per layer: fused q|k|v weight + bias (frozen copies) and the PreparedLinear handles of every projection."""
        layers = self.text_model.encoder.layers
        key = tuple(l.self_attn.q_proj.weight.data_ptr() for l in layers)
        if self._fused is not None and self._fused[0] == key:
            return self._fused[1]
        out = []
        for l in layers:
            a = l.self_attn
            w = nn.Parameter(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0).detach(), requires_grad=False)
            b = nn.Parameter(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0).detach(), requires_grad=False)
            out.append(dict(wqkv=w, bqkv=b, pqkv=Fn.PreparedLinear(w), pout=Fn.PreparedLinear(a.out_proj.weight),
                            pfc1=Fn.PreparedLinear(l.mlp.fc1.weight), pfc2=Fn.PreparedLinear(l.mlp.fc2.weight)))
        self._fused = (key, out)
        return out

    def forward(self, input_ids=None, inputs_embeds=None):
        tm = self.text_model
        if any(p.requires_grad for p in tm.encoder.parameters()):
            return super().forward(input_ids=input_ids, inputs_embeds=inputs_embeds)
        if inputs_embeds is None:
            inputs_embeds = tm.embeddings.token_embedding(input_ids)
        B, S, W = inputs_embeds.shape
        cfg = self.config
        H = cfg["num_heads"]
        DH = W // H
        act = Fn.quick_gelu if cfg["act"] == "quick_gelu" else Fn.gelu
        x = (inputs_embeds + tm.embeddings.position_embedding.weight[:S]).to(ops.ACT).reshape(B * S, W).contiguous()
        for l, f in zip(tm.encoder.layers, self._prepare()):
            n, xs = Fn.layer_norm_skip(x, l.layer_norm1.weight, l.layer_norm1.bias, l.layer_norm1.eps)
            qkv = Fn.linear(n, f["wqkv"], f["bqkv"], f["pqkv"])
            a = Fn.attention(qkv, None, B, H, S, S, DH, DH ** -0.5, causal=True)
            x = Fn.linear(a, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias, f["pout"], residual=xs)
            n, xs = Fn.layer_norm_skip(x, l.layer_norm2.weight, l.layer_norm2.bias, l.layer_norm2.eps)
            h = act(Fn.linear(n, l.mlp.fc1.weight, l.mlp.fc1.bias, f["pfc1"]))
            x = Fn.linear(h, l.mlp.fc2.weight, l.mlp.fc2.bias, f["pfc2"], residual=xs)
        fl = tm.final_layer_norm
        y = Fn.layer_norm(x, fl.weight, fl.bias, fl.eps)
        return (y.view(B, S, W),)
