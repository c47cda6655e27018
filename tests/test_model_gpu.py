"""-m gpu: whole-model parity on the real kernels (tiny configs, seconds): smoke step vs the CPU oracle, and the
kernel-driven VAE encoder vs its stock-torch twin."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_smoke_step_matches_oracle(hip_env):
    from e4t import smoke
    errs = smoke.run(torch.device("cuda:0"), verbose=False)
    assert errs["loss_diff"] < 2e-2


def test_native_vae_matches_torch(hip_env):
    from e4t.vae import VAEEncoder
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    vae = VAEEncoder(block_out_channels=(64, 128, 128)).requires_grad_(False).to(dev)
    x = torch.rand(2, 3, 64, 64, device=dev) * 2 - 1
    eps = torch.randn(2, 4, 16, 16, device=dev)
    z = vae.encode_sample(x, eps)
    z_ref = super(VAEEncoder, vae).encode_sample(x, eps)     # fp32 torch ops, same parameters
    rel = float((z - z_ref).norm() / z_ref.norm())
    assert rel < 2e-2, rel
