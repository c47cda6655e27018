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


def test_vit_restatement_matches_installed_transformers_clip_vision(emu_fp32):
    """[3P] leaf: open_clip is not installed, but transformers' CLIPVisionModel is — the same ViT (patch conv without bias, class
    token, learned positions, pre-LN, pre-norm blocks with erf-GELU MLP, post-LN on the pooled token only).  With the weights
    mapped by name, the oracle's VisionTransformer and (through the emulation) the native one must reproduce its
    `pooler_output` (= pooled) and `last_hidden_state[:, 1:]` (= the tokens before ln_post that E4T consumes)."""
    transformers = pytest.importorskip("transformers")
    from e4t.encoder import VisionTransformer
    cfg = transformers.CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=28, patch_size=14,
                                        hidden_act="gelu")
    torch.manual_seed(0)
    hf = transformers.CLIPVisionModel(cfg).eval()
    h = {k.replace("vision_model.", ""): v for k, v in hf.state_dict().items()}
    sd = {"conv1.weight": h["embeddings.patch_embedding.weight"], "class_embedding": h["embeddings.class_embedding"],
          "positional_embedding": h["embeddings.position_embedding.weight"], "ln_pre.weight": h["pre_layrnorm.weight"], "ln_pre.bias": h["pre_layrnorm.bias"],
          "ln_post.weight": h["post_layernorm.weight"], "ln_post.bias": h["post_layernorm.bias"]}
    for i in range(2):
        a, b = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        sd[b + "attn.in_proj_weight"] = torch.cat([h[a + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        sd[b + "attn.in_proj_bias"] = torch.cat([h[a + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        for src, dst in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            sd[b + dst + ".weight"], sd[b + dst + ".bias"] = h[a + src + ".weight"], h[a + src + ".bias"]
    x = torch.randn(2, 3, 28, 28, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = hf(pixel_values=x)
    vit_cfg = dict(image_size=28, patch_size=14, width=64, layers=2, heads=2, mlp_ratio=4.0)
    o = orc.VisionTransformer(**vit_cfg)
    o.load_state_dict(sd)
    with torch.no_grad():
        pooled, tokens = o(x)
    torch.testing.assert_close(pooled, want.pooler_output, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(tokens, want.last_hidden_state[:, 1:], rtol=1e-5, atol=1e-5)
    # the native ViT takes the E4T encoder's raw [-1, 1] image and resizes / normalises it itself: feed it the inverse of that
    # preprocessing at the ViT's own resolution, so that its patch embedding sees exactly x
    n = VisionTransformer(**vit_cfg)
    n.load_state_dict(sd)
    mean = torch.tensor(orc.CLIP_MEAN)[None, :, None, None]
    std = torch.tensor(orc.CLIP_STD)[None, :, None, None]
    raw = (x * std + mean) * 2 - 1
    with torch.no_grad():
        pooled_n, tokens_n = n(raw)
    torch.testing.assert_close(pooled_n.float(), want.pooler_output, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(tokens_n.float(), want.last_hidden_state[:, 1:], rtol=2e-4, atol=2e-4)

