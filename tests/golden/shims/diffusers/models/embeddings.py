"""[3P] leaves restated in oracle/e4t_oracle.py (Timesteps, TimestepEmbedding); the rest are never instantiated by SD configs"""
import torch
from torch import nn

import e4t_oracle as _orc


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return _orc.timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos, freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(_orc.TimestepEmbedding):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        assert act_fn == "silu" and out_dim is None and post_act_fn is None and cond_proj_dim is None
        super().__init__(in_channels, time_embed_dim)

    def forward(self, sample, condition=None):
        assert condition is None
        return super().forward(sample)


class _Never(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is not used by the Stable Diffusion configurations")


class GaussianFourierProjection(_Never):
    pass


class ImagePositionalEmbeddings(_Never):
    pass


class PatchEmbed(_Never):
    pass


class CombinedTimestepLabelEmbeddings(_Never):
    pass
