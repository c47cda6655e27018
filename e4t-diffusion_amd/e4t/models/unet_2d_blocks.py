"""The five UNet block types SD-1.x / SD-2.x instantiate (reference: e4t/models/unet_2d_blocks.py
:454-551 mid, :727-855 CrossAttnDown, :858-934 Down, :1697-1827 CrossAttnUp, :1830-1901 Up), as
sequencers over NHWC maps.  The up blocks never materialise ``torch.cat([hidden, skip], 1)``
(:1795, :1883): the pair is handed to the ResBlock, whose GroupNorm and shortcut GEMM read both sources."""
from __future__ import annotations

from torch import nn

from .resnet import Downsample2D, ResnetBlock2D, Upsample2D
from .transformer_2d import Transformer2DModel


def _res(cin, cout, temb, eps, groups):
    return ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups)


def _attn(heads, ch, ctx_dim, groups, linproj, only_cross, upcast):
    return Transformer2DModel(heads, ch // heads, in_channels=ch, num_layers=1, cross_attention_dim=ctx_dim, norm_num_groups=groups,
                              use_linear_projection=linproj, only_cross_attention=only_cross, upcast_attention=upcast)


class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, add_downsample=True, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps, resnet_groups)
                                      for i in range(num_layers)])
        self.attentions = nn.ModuleList([_attn(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups,
                                               use_linear_projection, only_cross_attention, upcast_attention) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, name="op")]) if add_downsample else None

    def forward_nhwc(self, m, temb_act, ctx, prefix=None):
        outs = ()
        for i, (r, a) in enumerate(zip(self.resnets, self.attentions)):
            if i == 0 and prefix is not None:      # shared prefix of the step's two UNet passes (unet_2d_condition.py)
                m_in = m
                m = a.forward_nhwc(prefix.reuse("res0", lambda: r.forward_nhwc(m_in, None, temb_act)), ctx, prefix)
            else:
                m = a.forward_nhwc(r.forward_nhwc(m, None, temb_act), ctx)
            outs += (m,)
        if self.downsamplers is not None:
            m = self.downsamplers[0].forward_nhwc(m)
            outs += (m,)
        return m, outs


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_downsample=True, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps, resnet_groups)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, name="op")]) if add_downsample else None

    def forward_nhwc(self, m, temb_act, ctx=None):
        outs = ()
        for r in self.resnets:
            m = r.forward_nhwc(m, None, temb_act)
            outs += (m,)
        if self.downsamplers is not None:
            m = self.downsamplers[0].forward_nhwc(m)
            outs += (m,)
        return m, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 cross_attention_dim=1280, use_linear_projection=False, upcast_attention=False, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups) for _ in range(2)])
        self.attentions = nn.ModuleList([_attn(attn_num_head_channels, in_channels, cross_attention_dim, resnet_groups,
                                               use_linear_projection, False, upcast_attention)])

    def forward_nhwc(self, m, temb_act, ctx):
        m = self.resnets[0].forward_nhwc(m, None, temb_act)
        m = self.attentions[0].forward_nhwc(m, ctx)
        return self.resnets[1].forward_nhwc(m, None, temb_act)


def _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups):
    res = []
    for i in range(num_layers):
        skip = in_channels if i == num_layers - 1 else out_channels
        rin = prev_output_channel if i == 0 else out_channels
        res.append(_res(rin + skip, out_channels, temb_channels, eps, groups))
    return nn.ModuleList(res)


class CrossAttnUpBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, add_upsample=True,
                 use_linear_projection=False, only_cross_attention=False, upcast_attention=False, **unused):
        super().__init__()
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps, resnet_groups)
        self.attentions = nn.ModuleList([_attn(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups,
                                               use_linear_projection, only_cross_attention, upcast_attention) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward_nhwc(self, m, skips, temb_act, ctx, upsample_size=None):
        for r, a in zip(self.resnets, self.attentions):
            s, skips = skips[-1], skips[:-1]
            m = a.forward_nhwc(r.forward_nhwc(m, s, temb_act), ctx)
        if self.upsamplers is not None:
            m = self.upsamplers[0].forward_nhwc(m, upsample_size)
        return m


class UpBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, add_upsample=True, **unused):
        super().__init__()
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps, resnet_groups)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward_nhwc(self, m, skips, temb_act, ctx=None, upsample_size=None):
        for r in self.resnets:
            s, skips = skips[-1], skips[:-1]
            m = r.forward_nhwc(m, s, temb_act)
        if self.upsamplers is not None:
            m = self.upsamplers[0].forward_nhwc(m, upsample_size)
        return m


DOWN_BLOCKS = {"CrossAttnDownBlock2D": CrossAttnDownBlock2D, "DownBlock2D": DownBlock2D}
UP_BLOCKS = {"CrossAttnUpBlock2D": CrossAttnUpBlock2D, "UpBlock2D": UpBlock2D}


def get_down_block(down_block_type, **kw):
    if down_block_type not in DOWN_BLOCKS:
        raise ValueError(f"{down_block_type} is not used by SD-1.x/2.x and is not built (SURVEY.md §2 #5)")
    return DOWN_BLOCKS[down_block_type](**kw)


def get_up_block(up_block_type, **kw):
    if up_block_type not in UP_BLOCKS:
        raise ValueError(f"{up_block_type} is not used by SD-1.x/2.x and is not built (SURVEY.md §2 #5)")
    return UP_BLOCKS[up_block_type](**kw)
