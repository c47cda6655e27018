"""kornia.geometry.resize / kornia.enhance.normalize as the reference calls them (e4t/encoder.py:131-139)"""
import torch.nn.functional as F


class geometry:
    @staticmethod
    def resize(x, size, interpolation="bilinear", align_corners=None, antialias=False):
        assert not antialias
        return F.interpolate(x, size=size, mode=interpolation, align_corners=align_corners)


class enhance:
    @staticmethod
    def normalize(x, mean, std):
        return (x - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1)
