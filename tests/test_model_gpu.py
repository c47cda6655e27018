"""-m gpu: whole-model parity on the real kernels (tiny configs, seconds): smoke step vs the CPU oracle, and the
kernel-driven VAE encoder / CLIP text encoder vs their stock-torch twins."""
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


def test_native_text_encoder_matches_torch(hip_env):
    """CLIP-L text encoder shapes (12 x 768, 12 heads, 77 tokens): forward and d/d inputs_embeds on the kernels vs fp32 torch."""
    from e4t.frozen import CLIPTextModel as TorchText
    from e4t.text import CLIPTextModel as NativeText
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    cfg = dict(vocab_size=1000, num_layers=4)
    ref = TorchText(**cfg).requires_grad_(False).to(dev)
    nat = NativeText(**cfg).requires_grad_(False).to(dev)
    nat.load_state_dict(ref.state_dict())
    B, S, W = 2, 77, 768
    e1 = (torch.randn(B, S, W, device=dev) * 0.3).requires_grad_(True)
    e2 = e1.detach().clone().requires_grad_(True)
    y1, y2 = nat(inputs_embeds=e1)[0].float(), ref(inputs_embeds=e2)[0]
    assert float((y1 - y2).norm() / y2.norm()) < 2e-2
    w = torch.randn(B, S, W, device=dev)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    assert float((e1.grad - e2.grad).norm() / e2.grad.norm()) < 3e-2
