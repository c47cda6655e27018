"""Pieces shared by tests/golden/make_golden_models.py and tests/test_reference_golden.py for the SAMPLING-LOOP fixture.
The reference's E4TEncoder hard-codes the 10880 pooled features of the full-size SD UNet (e4t/encoder.py:102), so the loop
fixture drives the reference pipeline with this small deterministic encoder instead (the real one is pinned separately)."""
import torch
from torch import nn


class StandInEncoder(nn.Module):
    """same call signature as E4TEncoder.forward(x, unet_down_block_samples) -> (B, dim)"""

    def __init__(self, n_features, dim):
        super().__init__()
        g = torch.Generator().manual_seed(77)
        self.w = nn.Parameter(torch.randn(dim, n_features + 3, generator=g) * 0.2, requires_grad=False)

    def forward(self, x, unet_down_block_samples):
        pooled = torch.cat([s.mean(dim=(2, 3)) for s in unet_down_block_samples] + [x.mean(dim=(2, 3))], dim=-1)
        return torch.tanh(pooled @ self.w.t())


TEXT_CFG = dict(vocab_size=100, hidden_size=12, num_layers=2, num_heads=2, intermediate_size=24, max_len=9, act="quick_gelu")
PROMPT = "a photo of *s"
