"""The check behind __graft_entry__.smoke() (kept under tests/: it needs the oracle, which product code must not import).
One small invocation of the whole hot path on cuda:0 through the HIP kernels, checked against
the CPU fp32 oracle (the oracle is only the checker here; see oracle/e4t_oracle.py header)."""
from __future__ import annotations

import torch

TINY_VIT = dict(image_size=28, patch_size=14, width=128, layers=2, heads=2, mlp_ratio=4.0)
BOC = (64, 128, 128, 128)
TEXT_CFG = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=128, max_len=9, act="quick_gelu")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def run(dev, verbose=True):
    import e4t_oracle as orc                      # checker only
    from e4t import ops
    from e4t.encoder import E4TEncoder
    from e4t.frozen import CLIPTextModel
    from e4t.text import CLIPTextModel as NativeCLIPTextModel
    from e4t.models.unet_2d_condition import UNet2DConditionModel
    from e4t.trainer import E4TTrainer

    assert isinstance(ops.backend(), ops.HipBackend), "smoke must run on the HIP backend"
    torch.manual_seed(0)
    cfg = orc.tiny_unet_config(ctx_dim=64)
    r_unet = orc.UNet2DConditionModel(**cfg)
    r_enc = orc.E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, vit_cfg=TINY_VIT)
    text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
    n_unet = UNet2DConditionModel(**cfg)
    n_unet.load_state_dict(r_unet.state_dict())
    n_enc = E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3)
    n_enc.load_state_dict(r_enc.state_dict())
    n_unet.to(dev); n_enc.to(dev)
    text_d = NativeCLIPTextModel(**TEXT_CFG).requires_grad_(False)      # text encoder on the HIP kernels too
    text_d.load_state_dict(text.state_dict())
    text_d.to(dev)

    B = 2
    g = torch.Generator().manual_seed(7)
    pixels = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    latents = torch.randn(B, 4, 16, 16, generator=g) * 0.18215
    noise = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([5, 700])
    ids = torch.randint(1, 99, (B, 9), generator=g)
    pidx = torch.tensor([2, 4])

    # ---- oracle (CPU fp32)
    for n, p in r_unet.named_parameters():
        p.requires_grad_("wo" in n)
    acp = orc.ddpm_alphas_cumprod()
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([11]))[0]
        ctx0 = text(input_ids=torch.zeros(1, 9, dtype=torch.long))[0]
        emb = text.get_input_embeddings()(ids)
    loss_r, ld_r, lr_r, aux = orc.e4t_losses(r_unet, r_enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds)[0], pixels, latents, noise,
                                             t, emb, pidx.tolist(), ctx0, class_embed, acp)
    loss_r.backward()

    # ---- native (GPU, bf16 HIP kernels)
    tr = E4TTrainer(n_unet, n_enc, text_d, vae=None, lr=1e-4, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=dev)
    d = lambda x: x.to(dev)
    with torch.no_grad():
        noisy = tr.add_noise(d(latents), d(noise), d(t))
        enc_n = n_unet(noisy, d(t), tr.ctx_for_e4t.expand(B, -1, -1), return_encoder_outputs=True)["down_block_samples"]
    errs = {}
    for i, (a, b) in enumerate(zip(enc_n, aux["enc"]["down_block_samples"])):
        errs[f"enc_map_{i}"] = rel(a, b)
    loss_n, ld_n, lr_n = tr.losses(d(pixels), d(latents), d(noise), d(t), d(ids), d(pidx))
    loss_n.backward()
    torch.cuda.synchronize()
    errs["loss_diff"] = abs(float(ld_n) - float(ld_r)) / abs(float(ld_r))
    errs["loss_reg"] = abs(float(lr_n) - float(lr_r)) / abs(float(lr_r))
    gr = dict(r_unet.named_parameters())
    worst_wo = 0.0
    for n, p in n_unet.named_parameters():
        if "wo" in n and p.numel() > 1:
            worst_wo = max(worst_wo, rel(p.grad, gr[n].grad))
    errs["worst_wo_grad"] = worst_wo
    ge = dict(r_enc.named_parameters())
    worst_e = 0.0
    for n, p in n_enc.named_parameters():
        if p.requires_grad:
            worst_e = max(worst_e, rel(p.grad, ge[n].grad))
    errs["worst_encoder_grad"] = worst_e
    if verbose:
        for k, v in errs.items():
            print(f"  smoke {k:<22s} rel_err={v:.3e}")
    # bf16 end-to-end tolerances (activations rounded to bf16 through ~60 layers; SURVEY.md §8c protocol)
    assert max(errs[f"enc_map_{i}"] for i in range(13)) < 3e-2, errs
    assert errs["loss_diff"] < 2e-2 and errs["loss_reg"] < 2e-2, errs
    assert errs["worst_wo_grad"] < 0.25 and errs["worst_encoder_grad"] < 0.25, errs
    # and the optimiser step runs
    tr.optimizer_step(); tr.zero_grad()
    torch.cuda.synchronize()
    print("smoke ok")
    return errs
