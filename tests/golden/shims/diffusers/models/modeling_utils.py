import torch
from torch import nn


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device
