"""CPU: the kernel-driven VAE encoder (e4t/vae.py) vs its stock-torch twin (tests/torch_twins.py) and vs the oracle's VAE,
same parameters, through the fp32 op emulation."""
import torch

import e4t_oracle as orc
import torch_twins
from test_unet_host_logic import emu_fp32  # noqa: F401


def to_oracle_keys(sd):
    out = {}
    for k, v in sd.items():
        k2 = k
        if k.startswith("encoder."):
            k2 = k[len("encoder."):]
            k2 = k2.replace("down_blocks.", "down.").replace("downsamplers.0.", "downsampler.")
            k2 = k2.replace("mid_block.resnets.0.", "mid_res1.").replace("mid_block.resnets.1.", "mid_res2.").replace("mid_block.attentions.0.", "mid_attn.")
        out[k2] = v
    return out


def test_native_vae_matches_torch_and_oracle(emu_fp32):
    from e4t.vae import VAEEncoder
    torch.manual_seed(0)
    boc = (64, 128, 128)
    nat = VAEEncoder(block_out_channels=boc).requires_grad_(False)
    ref = orc.VAEEncoder(block_out_channels=boc)
    ref.load_state_dict(to_oracle_keys(nat.state_dict()))
    x = torch.rand(2, 3, 32, 32) * 2 - 1
    eps = torch.randn(2, 4, 8, 8)
    z_nat = nat.encode_sample(x, eps)
    z_torch = torch_twins.vae_encode_sample(nat, x, eps)
    z_ref = ref.encode_sample(x, eps)
    torch.testing.assert_close(z_torch, z_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(z_nat, z_ref, rtol=2e-4, atol=2e-5)


def test_native_vae_decoder_matches_torch_and_oracle(emu_fp32):
    from e4t.vae import VAEDecoder
    torch.manual_seed(1)
    boc = (64, 128, 128)
    nat = VAEDecoder(block_out_channels=boc).requires_grad_(False)
    ref = orc.VAEDecoder(block_out_channels=boc)
    ref.load_state_dict(nat.state_dict())                    # same key names as the diffusers checkpoint
    z = torch.randn(2, 4, 8, 8) * 0.18215
    want = ref.decode_latents(z)
    torch.testing.assert_close(torch_twins.vae_decode_latents(nat, z), want, rtol=1e-4, atol=1e-5)
    got = nat.decode_latents(z)
    assert got.shape == (2, 32, 32, 3) and got.dtype == torch.float32
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(nat.decode(z), ref.decode(z), rtol=2e-4, atol=5e-5)
