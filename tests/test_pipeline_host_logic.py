"""CPU: the sampling pipeline — native StableDiffusionE4TPipeline (eager path, op emulation in fp32) against
the oracle's restatement of the reference loop (e4t_oracle.e4t_sample + DDIMScheduler + VAEDecoder), and the scheduler's
closed-form linear coefficients against the step-by-step formulation."""
import numpy as np
import pytest
import torch

import e4t_oracle as orc
from test_unet_host_logic import emu_fp32  # noqa: F401
from test_train_step_host_logic import build, TEXT_CFG
from word_tokenizer import WordTokenizer


def test_ddim_linear_coefficients_match_stepwise_form():
    from e4t.schedulers import DDIMScheduler
    g = torch.Generator().manual_seed(0)
    for pt in ("epsilon", "v_prediction"):
        ref = orc.DDIMScheduler(prediction_type=pt)
        nat = DDIMScheduler.stable_diffusion(prediction_type=pt)
        ref.set_timesteps(20)
        nat.set_timesteps(20)
        assert nat.timesteps.tolist() == ref.timesteps.tolist() and nat.timesteps[0] == 951 and nat.timesteps[-1] == 1
        for eta in (0.0, 0.7):
            for t in (951, 501, 1):
                x, e, n = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
                want = ref.step(e, t, x, eta=eta, variance_noise=n)
                cs, cp, cn = nat.coefficients(t, eta)
                torch.testing.assert_close(cs * x + cp * e + cn * n, want, rtol=2e-5, atol=2e-6)
    # the generic (clipping) formulation of the native class against the oracle's
    ref = orc.DDIMScheduler(clip_sample=True)
    nat = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True, set_alpha_to_one=False, steps_offset=1)
    ref.set_timesteps(10)
    nat.set_timesteps(10)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    torch.testing.assert_close(nat.step(e, 401, x).prev_sample, ref.step(e, 401, x), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        DDIMScheduler().coefficients(1)                     # diffusers' default config clips: not linear


def _pipeline(n_unet, n_enc, text, vae):
    from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
    from e4t.schedulers import DDIMScheduler
    tok = WordTokenizer()
    cfg = dict(placeholder_token="*s", domain_class_token="art", domain_embed_scale=0.1)
    return StableDiffusionE4TPipeline(vae=vae, text_encoder=text, tokenizer=tok, unet=n_unet, e4t_encoder=n_enc,
                                      scheduler=DDIMScheduler.stable_diffusion(), safety_checker=None, e4t_config=cfg), tok


@pytest.mark.parametrize("guidance", [1.0, 4.0])
def test_pipeline_matches_oracle_loop(emu_fp32, guidance):
    from e4t.vae import VAEDecoder
    r_unet, r_enc, n_unet, n_enc, text = build()
    torch.manual_seed(3)
    vae = VAEDecoder(block_out_channels=(64, 64)).requires_grad_(False)
    r_vae = orc.VAEDecoder(block_out_channels=(64, 64))
    r_vae.load_state_dict(vae.state_dict())
    pipe, tok = _pipeline(n_unet, n_enc, text, vae)
    assert pipe.vae_scale_factor == 2 and text.get_input_embeddings().weight.shape[0] == 101      # grown by the placeholder
    g = torch.Generator().manual_seed(5)
    image = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    lat0 = torch.randn(2, 4, 16, 16, generator=g)
    steps = 3
    out = pipe("a painting of *s", height=32, width=32, num_inference_steps=steps, guidance_scale=guidance, num_images_per_prompt=2,
               latents=lat0.clone(), image=image, output_type="latent", use_graph=False)
    # the oracle loop on the same weights
    ids = tok("a painting of *s", padding="max_length", max_length=9).input_ids
    idx = ids[0].tolist().index(tok.convert_tokens_to_ids("*s"))
    with torch.no_grad():
        emb = text.get_input_embeddings()(ids)
        ctx0 = text(tok("", padding="max_length", max_length=9).input_ids)[0]
        class_embed = text.get_input_embeddings()(torch.tensor([11]))
    want = orc.e4t_sample(r_unet, r_enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds)[0], orc.DDIMScheduler(), image, emb, idx,
                          ctx0, class_embed, lat0.clone(), num_inference_steps=steps, guidance_scale=guidance)
    torch.testing.assert_close(out.images, want, rtol=2e-3, atol=2e-4)
    # the unfused scheduler path (generic .step) gives the same latents
    out2 = pipe("a painting of *s", height=32, width=32, num_inference_steps=steps, guidance_scale=guidance, num_images_per_prompt=2,
                latents=lat0.clone(), image=image, output_type="latent", use_graph=False, eta=1e-12)
    torch.testing.assert_close(out2.images, out.images, rtol=1e-4, atol=1e-5)
    # decoded images: numpy NHWC in [0,1] / PIL
    imgs = pipe("a painting of *s", height=32, width=32, num_inference_steps=1, guidance_scale=guidance, latents=lat0[:1].clone(), image=image,
                output_type="np", use_graph=False).images
    assert imgs.shape == (1, 32, 32, 3) and imgs.dtype == np.float32 and 0.0 <= imgs.min() and imgs.max() <= 1.0
    lat1 = pipe("a painting of *s", height=32, width=32, num_inference_steps=1, guidance_scale=guidance, latents=lat0[:1].clone(), image=image,
                output_type="latent", use_graph=False).images
    np.testing.assert_allclose(imgs, r_vae.decode_latents(lat1).detach().numpy(), rtol=2e-3, atol=2e-4)
    pil = pipe("a painting of *s", height=32, width=32, num_inference_steps=1, latents=lat0[:1].clone(), image=image, use_graph=False).images
    assert pil[0].size == (32, 32)


def test_pipeline_argument_errors(emu_fp32):
    r_unet, r_enc, n_unet, n_enc, text = build()
    pipe, tok = _pipeline(n_unet, n_enc, text, vae=type("V", (), {"block_out_channels": (1, 1)})())
    img = torch.zeros(1, 3, 64, 64)
    with pytest.raises(ValueError, match="placeholder_token"):
        pipe("a painting", height=32, width=32, image=img, use_graph=False)
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe("a painting of *s", height=30, width=32, image=img, use_graph=False)
    with pytest.raises(AssertionError):
        pipe("a painting of *s", height=32, width=32, image=img, negative_prompt="x", use_graph=False)
    with pytest.raises(ValueError, match="already contains"):
        from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
        StableDiffusionE4TPipeline(vae=pipe.vae, text_encoder=text, tokenizer=tok, unet=n_unet, e4t_encoder=n_enc, scheduler=pipe.scheduler,
                                   e4t_config=dict(placeholder_token="*s", domain_class_token="art", domain_embed_scale=0.1))


@pytest.mark.parametrize("name", ["plms", "lms", "euler", "euler_ancestral", "dpm_solver++"])
def test_pipeline_runs_every_sampler(emu_fp32, name):
    """the generic path (scale_model_input / step(...).prev_sample / PLMS's extra timestep) end to end on the tiny models"""
    from e4t.schedulers import SCHEDULER_MAPPING
    from e4t.vae import VAEDecoder
    r_unet, r_enc, n_unet, n_enc, text = build()
    torch.manual_seed(3)
    vae = VAEDecoder(block_out_channels=(64, 64)).requires_grad_(False)
    pipe, tok = _pipeline(n_unet, n_enc, text, vae)
    pipe.scheduler = SCHEDULER_MAPPING[name].stable_diffusion()
    g = torch.Generator().manual_seed(5)
    image = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    calls = []
    out = pipe("a painting of *s", height=32, width=32, num_inference_steps=4, guidance_scale=3.0, image=image, output_type="latent",
               generator=torch.Generator().manual_seed(1), callback=lambda i, t, l: calls.append(float(t))).images
    assert out.shape == (1, 4, 16, 16) and torch.isfinite(out).all()
    assert len(calls) == (5 if name == "plms" else 4)
    with pytest.raises(ValueError, match="graph replay"):
        pipe("a painting of *s", height=32, width=32, num_inference_steps=2, image=image, use_graph=True)
