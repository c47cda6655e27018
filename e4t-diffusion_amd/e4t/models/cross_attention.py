"""CrossAttention with weight offsets + the native attention processor.

Mirrors the operator seam of the reference's e4t/models/cross_attention.py: ``CrossAttention`` keeps
``to_q/to_k/to_v/to_out/wo_q/wo_k/wo_v/heads/scale`` and the ``set_processor`` /
``processor(attn, hidden_states, encoder_hidden_states, attention_mask)`` plug-in API (:182-206);
``HipAttnProcessor`` drops in where ``AttnProcessor2_0`` / ``XFormersCrossAttnProcessor`` did
(:447-538) and runs:  [W o (1+WO())] (cached per step by the bank)  ->  fused q|k|v GEMM  ->
flash attention kernel  ->  out-projection GEMM with the block's residual add in its epilogue.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import functional as Fn
from ..weightoffsets import WeightOffsets, WOBank


class HipAttnProcessor:
    def __call__(self, attn: "CrossAttention", hidden_states, encoder_hidden_states=None, attention_mask=None, residual=None):
        assert attention_mask is None, "attention masks are never used on the E4T training path (unet_2d_blocks.py:817)"
        shp = hidden_states.shape
        B, T, d = shp if hidden_states.dim() == 3 else (attn._B, shp[0] // attn._B, shp[1])
        x = hidden_states.reshape(B * T, d)
        bank = attn._bank
        assert bank is not None, "CrossAttention used outside a UNet (no weight-offset bank registered)"
        token = bank._token
        H = attn.heads
        DH = attn.inner_dim // H
        if encoder_hidden_states is None:
            assert attn.is_self
            qkv = Fn.WOLinearFn.apply(x, token, bank.slots[attn._slot_qkv])
            o = Fn.attention(qkv, None, B, H, T, T, DH, attn.scale)
        else:
            assert not attn.is_self
            S = encoder_hidden_states.shape[1]
            ctx = encoder_hidden_states.reshape(B * S, encoder_hidden_states.shape[2])
            q = Fn.WOLinearFn.apply(x, token, bank.slots[attn._slot_q])
            kv = Fn.WOLinearFn.apply(ctx, token, bank.slots[attn._slot_kv])
            o = Fn.attention(q, kv, B, H, T, S, DH, attn.scale)
        res = residual.reshape(B * T, d) if residual is not None else None
        out = Fn.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, attn._out_prep, residual=res)
        return out.view(shp)


class CrossAttention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0, bias=False, upcast_attention: bool = False, upcast_softmax: bool = False,
                 processor=None, **unused):
        super().__init__()
        assert not bias and dropout == 0.0
        self.inner_dim = dim_head * heads
        self.is_self = cross_attention_dim is None
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        # upcast_attention / upcast_softmax (cross_attention.py:224-226,245-246; `stable-diffusion-2-1` @768 ships upcast_attention):
        # accepted and HONOURED BY CONSTRUCTION, not by a switch.  In the reference they are `.float()` casts in front of
        # torch.baddbmm / softmax of the math path; under the recipe this path replaces (`--mixed_precision bf16` = torch.autocast)
        # baddbmm is on autocast's cast-to-bf16 list, so the upcast query / key are cast straight back and Q.K^T runs with bf16
        # operands either way (checked: baddbmm(fp32, fp32) under autocast(bf16) returns bf16), and the SDPA / xFormers processors
        # (cross_attention.py:473-481,521-531) never look at the flags.  The kernel (csrc/attention.hip) contracts bf16 operands into
        # an fp32 accumulator, keeps the scores and the whole softmax in fp32 registers and never rounds S to bf16 — at least the
        # precision of either setting under autocast.  tests/test_configs_gpu.py::full_sd21 builds the UNet with upcast_attention=True
        # and holds it to the fp32 oracle.
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(cross_attention_dim, self.inner_dim, bias=False)
        self.to_v = nn.Linear(cross_attention_dim, self.inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim), nn.Dropout(dropout)])
        self.wo_q = WeightOffsets(query_dim, self.inner_dim)
        self.wo_k = WeightOffsets(cross_attention_dim, self.inner_dim)
        self.wo_v = WeightOffsets(cross_attention_dim, self.inner_dim)
        self._out_prep = Fn.PreparedLinear(self.to_out[0].weight)
        self._bank: Optional[WOBank] = None
        self._B = 1
        self.set_processor(processor or HipAttnProcessor())

    def register_bank(self, bank: WOBank):
        self._bank = bank
        if self.is_self:
            self._slot_qkv = bank.add([self.to_q, self.to_k, self.to_v], [self.wo_q, self.wo_k, self.wo_v])
        else:
            self._slot_q = bank.add([self.to_q], [self.wo_q])
            self._slot_kv = bank.add([self.to_k, self.to_v], [self.wo_k, self.wo_v])

    def set_use_memory_efficient_attention_xformers(self, use: bool, attention_op=None):
        # the reference's --enable_xformers_memory_efficient_attention switch: the native kernel already is
        # memory efficient (no T x S matrix in HBM); keep the call, select the native processor.
        self.set_processor(HipAttnProcessor())

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)
