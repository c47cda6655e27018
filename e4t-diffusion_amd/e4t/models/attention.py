"""BasicTransformerBlock / FeedForward / GEGLU on token matrices (reference: e4t/models/attention.py
:181-332, 335-384, 409-430).  Pre-LN block: LN -> self-attn (+res) -> LN -> cross-attn (+res) -> LN ->
GEGLU FF (+res); every residual add rides in the epilogue of the GEMM that produces the branch."""
from __future__ import annotations

from typing import Optional

from torch import nn

from .. import functional as Fn
from .cross_attention import CrossAttention


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)
        self._prep = Fn.PreparedLinear(self.proj.weight)

    def forward(self, x2d):
        return Fn.geglu(Fn.linear(x2d, self.proj.weight, self.proj.bias, self._prep))


class FeedForward(nn.Module):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0, activation_fn: str = "geglu", **unused):
        super().__init__()
        assert activation_fn == "geglu" and dropout == 0.0
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])
        self._prep = Fn.PreparedLinear(self.net[2].weight)

    def forward(self, x2d, residual=None):
        h = self.net[0](x2d)
        return Fn.linear(h, self.net[2].weight, self.net[2].bias, self._prep, residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu", attention_bias: bool = False,
                 only_cross_attention: bool = False, upcast_attention: bool = False, **unused):
        super().__init__()
        assert not only_cross_attention and cross_attention_dim is not None
        self.attn1 = CrossAttention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, bias=attention_bias,
                                    upcast_attention=upcast_attention)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.attn2 = CrossAttention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                                    dim_head=attention_head_dim, bias=attention_bias, upcast_attention=upcast_attention)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, prefix=None, **unused):
        """hidden_states: (B, T, dim) view of the token matrix; encoder_hidden_states: (B, S, ctx_dim) in ACT dtype.
        ``prefix``: the UNet's shared-prefix cache (unet_2d_condition.py) — the self-attention half of the very first
        block does not see the text context and is computed once for the two UNet passes of a step."""
        B, T, d = hidden_states.shape

        def self_attn_half():
            h = hidden_states.reshape(B * T, d)
            n, h = Fn.layer_norm_skip(h, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            return self.attn1(n.view(B, T, d), residual=h).reshape(B * T, d)
        shareable = prefix is not None and not getattr(self, "only_cross_attention", False)
        h = prefix.reuse("block0.attn1", self_attn_half) if shareable else self_attn_half()
        n, h = Fn.layer_norm_skip(h, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = self.attn2(n.view(B, T, d), encoder_hidden_states=encoder_hidden_states, residual=h).reshape(B * T, d)
        n, h = Fn.layer_norm_skip(h, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        h = self.ff(n, residual=h)
        return h.view(B, T, d)
