"""ResnetBlock2D / Downsample2D / Upsample2D on NHWC bf16 maps, driven by the HIP kernels.

These restate the [3P] diffusers==0.14.0 leaves the reference constructs at
e4t/models/unet_2d_blocks.py:481,522,760,804,881,900,1732,1774,1855,1872 (their source is not under
the reference tree): parameter names (``norm1, conv1, time_emb_proj, norm2, conv2, conv_shortcut``,
``conv``) are the SD checkpoint contract.

A ResBlock is 6 launches forward: GN-stats, GN-apply(+SiLU, reading the concat's two sources in
place), implicit-GEMM conv1 (+bias +time-embedding row bias), GN-stats, GN-apply, conv2 (+bias
+shortcut residual); the 1x1 shortcut is one GEMM whose A operand is the two concat sources.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _C
from .. import functional as Fn


class FMap:
    """An NHWC feature map: x is the [B*H*W, C] matrix; B, H, W its geometry."""
    __slots__ = ("x", "B", "H", "W")

    def __init__(self, x, B, H, W):
        self.x, self.B, self.H, self.W = x, B, H, W

    @property
    def C(self):
        return self.x.shape[1]

    def nchw(self):
        """NCHW-shaped (non-contiguous) view, for API compatibility with callers of the reference."""
        return self.x.view(self.B, self.H, self.W, self.C).permute(0, 3, 1, 2)

    @staticmethod
    def from_nchw(t, dtype):
        B, C, H, W = t.shape
        return FMap(t.permute(0, 2, 3, 1).reshape(B * H * W, C).to(dtype).contiguous(), B, H, W)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6, dropout=0.0,
                 time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True, **unused):
        super().__init__()
        out_channels = out_channels or in_channels
        assert time_embedding_norm == "default" and output_scale_factor == 1.0 and dropout == 0.0
        self.in_channels, self.out_channels, self.groups, self.eps = in_channels, out_channels, groups, eps
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self._p1, self._p2 = Fn.PreparedConv(self.conv1.weight), Fn.PreparedConv(self.conv2.weight)
        self._pt = Fn.PreparedLinear(self.time_emb_proj.weight) if self.time_emb_proj is not None else None
        self._ps = Fn.PreparedLinear(self.conv_shortcut.weight) if self.conv_shortcut is not None else None
        self._rb = None

    def forward_nhwc(self, m: FMap, skip: FMap = None, temb_act=None) -> FMap:
        """m (+ optional skip = the second torch.cat source); temb_act = silu(emb) as [B, temb] (ACT dtype)."""
        B, H, W = m.B, m.H, m.W
        geom = (B, H, W, H, W)
        x2 = skip.x if skip is not None else None
        # x1 / x2 come back as the operands for the shortcut: their gradient is folded into the GroupNorm backward
        h, x1, x2 = Fn.group_norm_skip(m.x, x2, self.norm1.weight, self.norm1.bias, B, H * W, self.groups, self.eps, True)
        rb = self._rb          # set by the UNet when all blocks' projections were evaluated in one GEMM (TimeEmbProjAllFn)
        if rb is None and temb_act is not None and self.time_emb_proj is not None:
            rb = Fn.linear(temb_act, self.time_emb_proj.weight, self.time_emb_proj.bias, self._pt, out_f32=True)
        h = Fn.conv3x3(h, self.conv1.weight, self.conv1.bias, self._p1, geom, rowbias=rb)
        h = Fn.group_norm(h, None, self.norm2.weight, self.norm2.bias, B, H * W, self.groups, self.eps, True)
        if self.conv_shortcut is not None:
            s = Fn.linear(x1, self.conv_shortcut.weight, self.conv_shortcut.bias, self._ps, x2=x2)
        else:
            assert skip is None
            s = x1
        y = Fn.conv3x3(h, self.conv2.weight, self.conv2.bias, self._p2, geom, residual=s)
        return FMap(y, B, H, W)

    def forward(self, x, temb=None):
        """Reference-compatible NCHW entry point (off the hot path)."""
        from .. import ops
        t = Fn.silu(temb.to(ops.ACT).contiguous()) if temb is not None else None
        return self.forward_nhwc(FMap.from_nchw(x, ops.ACT), None, t).nchw()


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1 and (out_channels or channels) == channels
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)
        self._p = Fn.PreparedConv(self.conv.weight)

    def forward_nhwc(self, m: FMap) -> FMap:
        Ho, Wo = (m.H - 1) // 2 + 1, (m.W - 1) // 2 + 1
        y = Fn.conv3x3(m.x, self.conv.weight, self.conv.bias, self._p, (m.B, m.H, m.W, Ho, Wo), mode=_C.CONV_S2)
        return FMap(y, m.B, Ho, Wo)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        assert use_conv and (out_channels or channels) == channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)
        self._p = Fn.PreparedConv(self.conv.weight)

    def forward_nhwc(self, m: FMap, output_size=None) -> FMap:
        assert output_size is None or tuple(output_size) == (2 * m.H, 2 * m.W), "only exact x2 nearest upsampling is built"
        y = Fn.conv3x3(m.x, self.conv.weight, self.conv.bias, self._p, (m.B, m.H, m.W, 2 * m.H, 2 * m.W), mode=_C.CONV_UP2)
        return FMap(y, m.B, 2 * m.H, 2 * m.W)
