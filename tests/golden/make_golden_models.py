"""Generates tests/golden/reference_*.pt by RUNNING THE REFERENCE'S OWN CODE (build container only: needs /root/reference).

    python tests/golden/make_golden_models.py

/root/reference/e4t/models/{cross_attention,attention,transformer_2d,unet_2d_blocks,unet_2d_condition}.py and
/root/reference/e4t/encoder.py are imported unmodified; `diffusers`, `kornia`, `open_clip` resolve to the stand-ins under
tests/golden/shims (see its README: only the third-party leaves come from this repository).  Each fixture holds the
configuration, the seeded state dict (the oracle uses the reference's parameter names, so it loads by key), the inputs, and
what the reference computed: outputs and parameter gradients.  tests/test_reference_golden.py replays them on the oracle.
Small on purpose (tiny widths, 8x8 latents): the whole set is ~3 MB.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "shims"), "/root/reference", os.path.join(ROOT, "oracle")]

import torch  # noqa: E402

torch.set_grad_enabled(True)
def pack(named):
    """{name: tensor} -> (one flat fp32 vector, [(name, shape)]): thousands of tiny tensors pickle to megabytes otherwise"""
    named = dict(named)
    return torch.cat([v.detach().reshape(-1).float() for v in named.values()]), [(k, tuple(v.shape)) for k, v in named.items()]


SD_MAP_CHANNELS = (320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280, 1280)      # sums to 10880 (encoder.py:102)


def unet_fixture():
    from e4t.models.unet_2d_condition import UNet2DConditionModel           # the reference's class
    cfg = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(8, 8, 16, 16), layers_per_block=2,
               cross_attention_dim=12, attention_head_dim=2, norm_num_groups=2, norm_eps=1e-5,
               down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
               up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"))
    out = {}
    for name, extra in (("sd1", {}), ("sd2", dict(use_linear_projection=True, attention_head_dim=(1, 2, 2, 4)))):
        torch.manual_seed(0)
        c = dict(cfg, **extra)
        unet = UNet2DConditionModel(**c)
        g = torch.Generator().manual_seed(1)
        sample = torch.randn(2, 4, 8, 8, generator=g)
        t = torch.tensor([3, 977])
        ctx = torch.randn(2, 5, 12, generator=g)
        G = torch.randn(2, 4, 8, 8, generator=g)
        enc = unet(sample, t, ctx, return_encoder_outputs=True)["down_block_samples"]
        full = unet(sample, t, ctx).sample
        scalar_t = unet(sample[:1], 500, ctx[:1]).sample                     # python-int timestep path (:446-455)
        (full * G).sum().backward()
        grads = {n: p.grad.clone() for n, p in unet.named_parameters() if "wo" in n}
        assert len(enc) == 13 and len(grads) == 96 * 9 and all(v.abs().sum() > 0 for v in grads.values())
        out[name] = dict(config=c, state_dict=pack(unet.state_dict()), sample=sample, timestep=t, ctx=ctx,
                         G=G, down_block_samples=[e.detach() for e in enc], out=full.detach(), out_scalar_t=scalar_t.detach(), wo_grads=pack(grads),
                         grad_conv_in=unet.conv_in.weight.grad.clone())
    return out


def attention_fixture():
    from e4t.models.cross_attention import AttnProcessor2_0, CrossAttention, CrossAttnProcessor
    out = {}
    for name, cad in (("self", None), ("cross", 24)):
        torch.manual_seed(2)
        attn = CrossAttention(query_dim=16, cross_attention_dim=cad, heads=2, dim_head=8)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 7, 16, generator=g)
        ctx = torch.randn(2, 5, 24, generator=g) if cad else None
        res = {}
        for pname, proc in (("math", CrossAttnProcessor()), ("sdpa", AttnProcessor2_0())):
            attn.set_processor(proc)
            attn.zero_grad()
            y = attn(x, encoder_hidden_states=ctx)
            y.square().sum().backward()
            res[pname] = dict(out=y.detach(), grads=pack({n: p.grad.clone() for n, p in attn.named_parameters()}))
        out[name] = dict(kwargs=dict(query_dim=16, cross_attention_dim=cad, heads=2, dim_head=8),
                         state_dict=pack(attn.state_dict()), x=x, ctx=ctx, **res)
    return out


def encoder_fixture():
    from e4t.encoder import E4TEncoder                                       # the reference's class
    torch.manual_seed(4)
    enc = E4TEncoder(word_embedding_dim=24, arch="ViT-golden-test", version="none", n_odd_layers=9, freeze_clip_vision=False)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 40, 48, generator=g) * 2 - 1
    maps = [torch.randn(2, c, 1 + (i % 2), 2, generator=g) for i, c in enumerate(SD_MAP_CHANNELS)]
    y = enc(x, tuple(maps))
    y.square().sum().backward()
    grads = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    pre = enc.preprocess(x).detach()[:, :, ::16, ::16].clone()          # a 14x14 sample of the 224x224 CLIP input
    return dict(kwargs=dict(word_embedding_dim=24, n_odd_layers=9), state_dict=pack(enc.state_dict()),
                x=x, maps=maps, out=y.detach(), grads=pack(grads), preprocessed=pre)


if __name__ == "__main__":
    import open_clip
    open_clip.TEST_ARCHS["ViT-golden-test"] = dict(image_size=224, patch_size=56, width=8, layers=2, heads=2, mlp_ratio=2.0)
    for name, fn in (("unet", unet_fixture), ("attention", attention_fixture), ("encoder", encoder_fixture)):
        blob = fn()
        path = os.path.join(HERE, f"reference_{name}.pt")
        torch.save(blob, path)
        print(f"{path}: {os.path.getsize(path) / 1e6:.2f} MB")
