"""Generation throughput of the E4T pipeline on one MI355X (SD-1.4 shapes, 512 px, DDIM, CFG 7.5): eager launch loop vs
hipGraph replay of the denoising step, plus the VAE decode.  Random-init weights, word-level stand-in tokenizer."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from bench import build_models
from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
from e4t.schedulers import DDIMScheduler
from e4t.vae import VAEDecoder
from word_tokenizer import WordTokenizer

dev = torch.device("cuda:0")
steps = int(os.environ.get("STEPS", "50"))
unet, enc, text, _ = build_models(dev, "sd14", seed=0)
unet.requires_grad_(False); enc.requires_grad_(False)
with torch.device(dev):
    vae = VAEDecoder().requires_grad_(False)
tok = WordTokenizer(base_size=49408, model_max_length=77)
pipe = StableDiffusionE4TPipeline(vae=vae, text_encoder=text, tokenizer=tok, unet=unet, e4t_encoder=enc, scheduler=DDIMScheduler.stable_diffusion(),
                                  e4t_config=dict(placeholder_token="*s", domain_class_token="art", domain_embed_scale=0.1), already_added_placeholder_token=False)
image = torch.rand(1, 3, 512, 512) * 2 - 1
for n in (1, 4):
    for graph in (False, True):
        kw = dict(num_inference_steps=steps, guidance_scale=7.5, num_images_per_prompt=n, image=image, output_type="np", use_graph=graph)
        pipe("a painting of *s", **dict(kw, num_inference_steps=2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe("a painting of *s", **kw).images
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"images/call={n} graph={graph}: {dt:.2f} s for {steps} steps ({dt/steps*1e3:.1f} ms/step incl. capture+decode) -> {n/dt:.2f} img/s; out {out.shape}", flush=True)
z = torch.randn(4, 4, 64, 64, device=dev) * 0.18215
vae.decode_latents(z); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    vae.decode_latents(z)
torch.cuda.synchronize(); print(f"VAE decode B=4 512px: {(time.perf_counter()-t0)/5*1e3:.1f} ms")
