from torch import nn


class DualTransformer2DModel(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("DualTransformer2DModel is not used by the Stable Diffusion configurations")
