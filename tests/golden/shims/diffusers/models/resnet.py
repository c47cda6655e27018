"""[3P diffusers 0.14] ResnetBlock2D / Downsample2D / Upsample2D with their constructor signatures, on the oracle's restatement"""
from torch import nn

import e4t_oracle as _orc


class ResnetBlock2D(_orc.ResnetBlock2D):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32, groups_out=None,
                 pre_norm=True, eps=1e-6, non_linearity="swish", time_embedding_norm="default", kernel=None, output_scale_factor=1.0,
                 use_in_shortcut=None, up=False, down=False):
        assert dropout == 0.0 and pre_norm and non_linearity in ("swish", "silu") and time_embedding_norm == "default"
        assert output_scale_factor == 1.0 and not up and not down and groups_out is None and kernel is None
        super().__init__(in_channels, out_channels or in_channels, temb_channels or 0, groups, eps)


class Downsample2D(_orc.Downsample2D):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        assert use_conv and (out_channels is None or out_channels == channels) and name == "op"
        super().__init__(channels, padding)


class Upsample2D(_orc.Upsample2D):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        assert use_conv and not use_conv_transpose and (out_channels is None or out_channels == channels)
        super().__init__(channels)

    def forward(self, hidden_states, output_size=None):
        return super().forward(hidden_states, output_size)


class _Never(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is not used by the Stable Diffusion configurations")


class FirDownsample2D(_Never):
    pass


class FirUpsample2D(_Never):
    pass


class KDownsample2D(_Never):
    pass


class KUpsample2D(_Never):
    pass
