"""-m gpu: whole-model parity on the real kernels (tiny configs, seconds): smoke step vs the CPU oracle, and the
kernel-driven VAE encoder / decoder / CLIP text encoder vs their stock-torch twins; the sampling pipeline eager vs hipGraph replay."""
import pytest
import torch

import torch_twins

pytestmark = pytest.mark.gpu


def test_smoke_step_matches_oracle(hip_env):
    import smoke_step as smoke
    errs = smoke.run(torch.device("cuda:0"), verbose=False)
    assert errs["losses"] < 2e-2


def test_native_vae_matches_torch(hip_env):
    from e4t.vae import VAEEncoder
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    vae = VAEEncoder(block_out_channels=(64, 128, 128)).requires_grad_(False).to(dev)
    x = torch.rand(2, 3, 64, 64, device=dev) * 2 - 1
    eps = torch.randn(2, 4, 16, 16, device=dev)
    z = vae.encode_sample(x, eps)
    z_ref = torch_twins.vae_encode_sample(vae, x, eps)       # fp32 torch ops, same parameters
    rel = float((z - z_ref).norm() / z_ref.norm())
    assert rel < 2e-2, rel


def test_native_text_encoder_matches_torch(hip_env):
    """CLIP-L text encoder shapes (12 x 768, 12 heads, 77 tokens): forward and d/d inputs_embeds on the kernels vs fp32 torch."""
    from torch_twins import CLIPTextModel as TorchText
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


def test_native_vae_decoder_matches_torch(hip_env):
    from e4t.vae import VAEDecoder
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    vae = VAEDecoder(block_out_channels=(64, 128, 128)).requires_grad_(False).to(dev)
    z = torch.randn(2, 4, 16, 16, device=dev) * 0.18215
    img = vae.decode_latents(z)
    ref = torch_twins.vae_decode_latents(vae, z)             # fp32 torch ops, same parameters
    assert img.shape == (2, 64, 64, 3) and img.dtype == torch.float32
    rel = float((img - ref).norm() / ref.norm())
    assert rel < 2e-2, rel


def test_pipeline_graph_replay_equals_eager(hip_env):
    """tiny models on the real kernels: the hipGraph-replayed denoising loop must reproduce the eager loop (same kernels, same
    order -> bit-identical latents), and both stay close to the fp32 CPU oracle loop."""
    import e4t_oracle as orc
    from word_tokenizer import WordTokenizer
    from test_train_step_host_logic import build
    from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
    from e4t.schedulers import DDIMScheduler
    from e4t.text import CLIPTextModel
    from e4t.vae import VAEDecoder
    dev = torch.device("cuda:0")
    r_unet, r_enc, n_unet, n_enc, text_t = build()
    text = CLIPTextModel(**text_t.config).requires_grad_(False)
    text.load_state_dict(text_t.state_dict())
    vae = VAEDecoder(block_out_channels=(64, 64)).requires_grad_(False)
    tok = WordTokenizer()
    pipe = StableDiffusionE4TPipeline(vae=vae, text_encoder=text, tokenizer=tok, unet=n_unet, e4t_encoder=n_enc,
                                      scheduler=DDIMScheduler.stable_diffusion(), safety_checker=None,
                                      e4t_config=dict(placeholder_token="*s", domain_class_token="art", domain_embed_scale=0.1)).to(dev)
    text_t.resize_token_embeddings(len(tok))
    text_t.load_state_dict(text.state_dict())
    g = torch.Generator().manual_seed(5)
    image = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    lat0 = torch.randn(2, 4, 16, 16, generator=g)
    kw = dict(height=32, width=32, num_inference_steps=4, guidance_scale=5.0, num_images_per_prompt=2, image=image, output_type="latent")
    eager = pipe("a painting of *s", latents=lat0.clone(), use_graph=False, **kw).images
    graph = pipe("a painting of *s", latents=lat0.clone(), use_graph=True, **kw).images
    assert torch.equal(eager, graph)
    ids = tok("a painting of *s", padding="max_length", max_length=9).input_ids
    idx = ids[0].tolist().index(tok.convert_tokens_to_ids("*s"))
    with torch.no_grad():
        emb = text_t.get_input_embeddings()(ids)
        ctx0 = text_t(tok("", padding="max_length", max_length=9).input_ids)[0]
        class_embed = text_t.get_input_embeddings()(torch.tensor([11]))
    want = orc.e4t_sample(r_unet, r_enc, lambda inputs_embeds: text_t(inputs_embeds=inputs_embeds)[0], orc.DDIMScheduler(), image, emb, idx,
                          ctx0, class_embed, lat0.clone(), num_inference_steps=4, guidance_scale=5.0)
    rel = float((eager.cpu() - want).norm() / want.norm())
    assert rel < 3e-2, rel
    img = pipe("a painting of *s", latents=lat0[:1].clone(), height=32, width=32, num_inference_steps=2, image=image).images
    assert img[0].size == (32, 32)


def test_training_step_is_bitwise_deterministic(hip_env):
    """SURVEY §8c parity protocol: run-to-run determinism of the native path.  No kernel uses atomics (split-K, column
    reductions and the attention backward all reduce in a fixed order), so two runs from the same state must agree bit
    for bit — losses and every trained parameter after two optimiser steps."""
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B = 2
    batches = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g) * 0.18215,
                torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randint(1, 99, (B, 9), generator=g))
               for _ in range(2)]
    pidx = torch.tensor([2, 4], device=dev)

    def run():
        _, _, n_unet, n_enc, text_t = build(seed=0)
        text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
        text.load_state_dict(text_t.state_dict())
        n_unet.to(dev), n_enc.to(dev), text.to(dev)
        tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev), device=dev)
        losses = []
        for px, lat, noise, t, ids in batches:
            out = tr.train_step(px.to(dev), ids.to(dev), pidx, noise=noise.to(dev), timesteps=t.to(dev), latents=lat.to(dev))
            losses.append(torch.stack([o.detach().float() for o in out]).cpu())
        torch.cuda.synchronize()
        return torch.stack(losses), tr.flat.data.detach().cpu().clone()

    l0, p0 = run()
    l1, p1 = run()
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())


@pytest.mark.parametrize("mode", ["vit", "vae", "vit+vae"])
def test_next_batch_prefetch_is_result_preserving(hip_env, mode):
    """E4TTrainer.prefetch(): the frozen CLIP-ViT (and VAE encoder) of batch i+1 run on the side stream under step i — same
    kernels on the same inputs, so losses and every trained parameter after three steps must equal the un-prefetched run bit for
    bit (the VAE's sampling noise is handed in, so the comparison does not depend on the order of torch's random draws)."""
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    from e4t.vae import VAEEncoder
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    B = 2
    batches = [(torch.rand(B, 3, 128, 128, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g),
                torch.randint(1, 99, (B, 9), generator=g), torch.randn(B, 4, 16, 16, generator=g)) for _ in range(3)]
    pidx = torch.tensor([2, 4], device=dev)

    def run(mode):
        _, _, n_unet, n_enc, text_t = build(seed=0)
        text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
        text.load_state_dict(text_t.state_dict())
        torch.manual_seed(11)
        vae = VAEEncoder(block_out_channels=(64, 128, 128, 128)).requires_grad_(False)
        n_unet.to(dev), n_enc.to(dev), text.to(dev), vae.to(dev)
        tr = E4TTrainer(n_unet, n_enc, text, vae=vae, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev), device=dev)
        tr.prefetch_mode = mode
        dbatches = [tuple(t.to(dev) for t in b) for b in batches]
        losses, used = [], 0
        for i, (px, noise, t, ids, eps) in enumerate(dbatches):
            if i + 1 < len(dbatches):
                tr.prefetch(dbatches[i + 1][0], vae_eps=dbatches[i + 1][4])
            took = bool(tr._pref.get(id(px))) and mode != "0"
            used += took
            # a step whose latents were prefetched must not be handed vae_eps (that asks for its own VAE pass)
            out = tr.train_step(px, ids, pidx, noise=noise, timesteps=t, vae_eps=None if (took and "vae" in mode) else eps)
            losses.append(torch.stack([o.detach().float() for o in out]).cpu())
        torch.cuda.synchronize()
        return torch.stack(losses), tr.flat.data.detach().cpu().clone(), used

    l0, p0, u0 = run("0")
    l1, p1, u1 = run(mode)
    assert u0 == 0 and u1 == len(batches) - 1          # every step after the first consumed what the step before it prefetched
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())


@pytest.mark.parametrize("given", [True, False], ids=["inputs_given", "draws_inside"])
def test_step_graph_replay_equals_eager(hip_env, given):
    """E4TTrainer.enable_step_graph(): the whole training step (VAE encode, both UNet passes, encoder, text encoder, backward, AdamW,
    zero-grad) captured into one HIP graph and replayed.  Same kernels in the same order on the same inputs: losses and every trained
    parameter after four steps (1 eager warm-up, 1 capture + replay, 2 replays) must equal the eager run bit for bit — with the noise /
    timesteps handed in, and with torch's graph-safe Philox draws inside the graph."""
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    from e4t.vae import VAEEncoder
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    B = 2
    batches = [(torch.rand(B, 3, 128, 128, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g),
                torch.randint(1, 99, (B, 9), generator=g)) for _ in range(4)]
    pidx = torch.tensor([2, 4], device=dev)

    def run(graph):
        _, _, n_unet, n_enc, text_t = build(seed=0)
        text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
        text.load_state_dict(text_t.state_dict())
        torch.manual_seed(11)
        vae = VAEEncoder(block_out_channels=(64, 128, 128, 128)).requires_grad_(False)
        n_unet.to(dev), n_enc.to(dev), text.to(dev), vae.to(dev)
        tr = E4TTrainer(n_unet, n_enc, text, vae=vae, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev), device=dev)
        assert tr.enable_step_graph(True)
        tr._step_graph_on = graph                  # the eager leg keeps the device-side AdamW scalars, launches from the host
        torch.manual_seed(5)
        losses = []
        for px, noise, t, ids in batches:
            kw = dict(noise=noise.to(dev), timesteps=t.to(dev)) if given else {}
            out = tr.train_step(px.to(dev), ids.to(dev), pidx, **kw)
            losses.append(torch.stack([o.detach().float() for o in out]).cpu())
        torch.cuda.synchronize()
        return torch.stack(losses), tr.flat.data.detach().cpu().clone(), len(tr._step_graphs)

    l0, p0, n0 = run(False)
    l1, p1, n1 = run(True)
    assert n0 == 0 and n1 == 1
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())


def test_step_graph_replays_queued_ahead_keep_their_own_adamw_scalars(hip_env):
    """Round-4 advisor finding: with graph replay the host enqueues a step far faster than the GPU runs it, so several steps' worth of
    AdamW scalars (lr, 1 - beta1^t, sqrt(1 - beta2^t)) are in flight at once; each queued update must see the values of ITS step.  The
    device is held busy (torch.cuda._sleep) while six replays are enqueued with no synchronisation in between; losses and parameters
    must equal the eager run that synchronises every step, bit for bit (bc1 is 0.1 at t = 1 and 0.41 at t = 5: a mixed-up step is an
    O(1) difference in the update)."""
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(13)
    B, steps = 2, 8
    batches = [(torch.rand(B, 3, 128, 128, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g), torch.randn(B, 4, 16, 16, generator=g),
                torch.randint(0, 1000, (B,), generator=g), torch.randint(1, 99, (B, 9), generator=g)) for _ in range(steps)]
    pidx = torch.tensor([2, 4], device=dev)

    def run(graph):
        _, _, n_unet, n_enc, text_t = build(seed=0)
        text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
        text.load_state_dict(text_t.state_dict())
        n_unet.to(dev), n_enc.to(dev), text.to(dev)
        tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev), device=dev)
        if graph:
            assert tr.enable_step_graph(True)
        dbatches = [tuple(t.to(dev) for t in b) for b in batches]
        losses = []
        for i, (px, lat, noise, t, ids) in enumerate(dbatches):
            if graph and i == 2:
                torch.cuda.synchronize()
                torch.cuda._sleep(int(1.5e9))          # ~0.7 s of busy device: every remaining replay is enqueued behind it
            out = tr.train_step(px, ids, pidx, noise=noise, timesteps=t, latents=lat)
            losses.append(torch.stack([o.detach().float() for o in out]))
            if not graph:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), tr.flat.data.detach().cpu().clone(), len(tr._step_graphs)

    l0, p0, n0 = run(False)
    l1, p1, n1 = run(True)
    assert n0 == 0 and n1 == 1
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
