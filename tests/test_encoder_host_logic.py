"""CPU: native E4TEncoder graph (ViT tower wiring, pooled-feature path, restructured 129-slot head and its
hand-written backward that writes parameter grads in place) vs the fp32 oracle, through the op emulation."""
import pytest
import torch

import e4t_oracle as orc
from test_unet_host_logic import emu_fp32  # noqa: F401  (fixture)

TINY_VIT = dict(image_size=28, patch_size=14, width=128, layers=2, heads=2, mlp_ratio=4.0)
BOC = (64, 128, 128, 128)


def make_maps(B, g):
    chans = [BOC[0], BOC[0], BOC[0], BOC[0], BOC[1], BOC[1], BOC[1], BOC[2], BOC[2], BOC[2], BOC[3], BOC[3], BOC[3]]
    sizes = [16, 16, 16, 8, 8, 8, 4, 4, 4, 2, 2, 2, 2]
    return [torch.randn(B, c, s, s, generator=g) for c, s in zip(chans, sizes)]


@pytest.mark.parametrize("unfreeze", [False, True])
def test_encoder_matches_oracle(emu_fp32, unfreeze):
    from e4t.encoder import E4TEncoder
    torch.manual_seed(0)
    ref = orc.E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, vit_cfg=TINY_VIT, freeze_clip_vision=not unfreeze)
    nat = E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3, freeze_clip_vision=not unfreeze)
    assert set(ref.state_dict()) == set(nat.state_dict()), set(ref.state_dict()) ^ set(nat.state_dict())
    nat.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(2)
    B = 3
    x = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    maps_r = [m.requires_grad_(True) for m in make_maps(B, g)]
    # native consumes NCHW-shaped views of NHWC storage, as the native UNet returns them
    maps_n = [m.detach().permute(0, 2, 3, 1).contiguous().requires_grad_(True) for m in maps_r]
    w = torch.randn(B, 64, generator=g)
    out_r = ref(x, maps_r)
    (out_r * w).sum().backward()
    out_n = nat(x, tuple(m.permute(0, 3, 1, 2) for m in maps_n))
    torch.testing.assert_close(out_n, out_r, rtol=5e-4, atol=5e-4)
    (out_n * w).sum().backward()
    for a, b in zip(maps_n, maps_r):
        torch.testing.assert_close(a.grad.permute(0, 3, 1, 2), b.grad, rtol=5e-3, atol=1e-5)
    gr = dict(ref.named_parameters())
    n_checked = 0
    for n, p in nat.named_parameters():
        if not p.requires_grad:
            assert gr[n].grad is None
            continue
        assert p.grad is not None, n
        torch.testing.assert_close(p.grad, gr[n].grad, rtol=5e-3, atol=2e-5, msg=lambda m, n=n: f"{n}: {m}")
        n_checked += 1
    assert n_checked >= 4 + 2 + 6 + 2
    # a second backward accumulates into the same persistent .grad storage
    out_n = nat(x, tuple(m.permute(0, 3, 1, 2) for m in maps_n))
    (out_n * w).sum().backward()
    torch.testing.assert_close(nat.first_linears[1].weight.grad, 2 * gr["first_linears.1.weight"].grad, rtol=5e-3, atol=4e-5)
    torch.testing.assert_close(nat.feature_linear.bias.grad, 2 * gr["feature_linear.bias"].grad, rtol=5e-3, atol=4e-5)
