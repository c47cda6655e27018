from dataclasses import dataclass

import torch

from ..utils import BaseOutput


@dataclass
class UNet2DConditionOutput(BaseOutput):
    sample: torch.FloatTensor = None
