"""Transformer2DModel, continuous-input branch (reference: e4t/models/transformer_2d.py:146-153,
249-286).  On NHWC maps the reference's two permutes (:257, :280) vanish: GroupNorm -> proj_in GEMM ->
transformer block -> proj_out GEMM with the map-level residual fused into its epilogue."""
from __future__ import annotations

from typing import Optional

from torch import nn

from .. import functional as Fn
from .attention import BasicTransformerBlock
from .resnet import FMap


class Transformer2DModel(nn.Module):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88, in_channels: Optional[int] = None,
                 num_layers: int = 1, dropout: float = 0.0, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 use_linear_projection: bool = False, only_cross_attention: bool = False, upcast_attention: bool = False, **unused):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.use_linear_projection, self.groups = use_linear_projection, norm_num_groups
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, only_cross_attention=only_cross_attention,
                                  upcast_attention=upcast_attention) for _ in range(num_layers)])
        # proj_out's output is the next ResBlock's GroupNorm input: its GEMM epilogue leaves the column statistics behind
        self._pin, self._pout = Fn.PreparedLinear(self.proj_in.weight), Fn.PreparedLinear(self.proj_out.weight, colstats=True)

    def forward_nhwc(self, m: FMap, ctx, prefix=None) -> FMap:
        B, HW = m.B, m.H * m.W

        def project_in():      # -> (proj_in(GN(x)), x as the residual operand of proj_out)
            g, res, _ = Fn.group_norm_skip(m.x, None, self.norm.weight, self.norm.bias, B, HW, self.groups, 1e-6, False)
            return Fn.linear(g, self.proj_in.weight, self.proj_in.bias, self._pin), res     # 1x1 conv == per-pixel GEMM
        h, res = prefix.reuse("attn0.proj_in", project_in) if prefix is not None else project_in()
        d = h.shape[1]
        h = h.view(B, HW, d)
        for i, blk in enumerate(self.transformer_blocks):
            h = blk(h, encoder_hidden_states=ctx, prefix=prefix if i == 0 else None)
        y = Fn.linear(h.reshape(B * HW, d), self.proj_out.weight, self.proj_out.bias, self._pout, residual=res)
        return FMap(y, B, m.H, m.W)
