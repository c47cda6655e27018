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


def deterministic_fill(module, salt=0):
    """Overwrite every parameter with values that depend only on (salt, parameter name, shape): lets a generator (reference
    model) and a test (oracle / native model with the same state-dict keys) hold identical weights WITHOUT storing them."""
    import zlib
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            g = torch.Generator().manual_seed(zlib.crc32(f"{salt}:{name}".encode()))
            x = torch.randn(p.shape, generator=g)
            if name.endswith("norm.weight") or ".norm" in name and name.endswith(".weight") or "conv_norm_out.weight" in name:
                p.copy_(1.0 + 0.1 * x)
            elif p.dim() == 1:
                p.copy_(0.1 * x)
            elif ".wo_" in name:                         # keep the offsets moderate: Delta = O(0.2)
                p.copy_(x * (0.6 / max(1, p.shape[-1]) ** 0.5))
            else:
                fan_in = p[0].numel()
                p.copy_(x * (1.0 / fan_in) ** 0.5)
    return module
