"""E4T inference on MI355X — same command line as the reference's inference.py (:34-50).

    python inference.py --pretrained_model_name_or_path <dir> --image_path_or_url in.png --prompt "a photo of *s"

<dir> is a directory written by pretrain_e4t.py / tuning_e4t.py (config.json, weight_offsets.pt | unet.pt, encoder.pt,
optionally text_encoder.pt); the base Stable Diffusion weights are read from the directory named in its config
(state dicts unet.pt / vae.pt / text_encoder.pt and a tokenizer/ folder) — there is no hub access here.  With
--random_init everything is randomly initialised and an offline whitespace tokenizer is used (smoke runs, benchmarks).
All six samplers of the reference are available; DDIM additionally has the fused, graph-replayed update.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "e4t-diffusion_amd"))

import torch  # noqa: E402
from PIL import Image  # noqa: E402


def image_grid(imgs, rows, cols):
    assert len(imgs) == rows * cols
    w, h = imgs[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, img in enumerate(imgs):
        grid.paste(img, box=(i % cols * w, i // cols * h))
    return grid


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--image_path_or_url", type=str, help="path to the input image")
    p.add_argument("--pretrained_model_name_or_path", type=str, help="model dir including config.json, encoder.pt, weight_offsets.pt")
    p.add_argument("--prompt", type=str, nargs="?", default="a photo of *s", help="the prompt to render (several joined by '::')")
    p.add_argument("--num_inference_steps", type=int, default=50)
    p.add_argument("--guidance_scale", type=float, default=1.0)
    p.add_argument("--num_images_per_prompt", type=int, default=1)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--scheduler_type", type=str, choices=["ddim", "plms", "lms", "euler", "euler_ancestral", "dpm_solver++"], default="ddim")
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--random_init", action="store_true", help="random weights + offline tokenizer (no checkpoint needed)")
    p.add_argument("--unet_variant", type=str, default="sd14", choices=["sd14", "sd21"])
    p.add_argument("--output", type=str, default="grid.png")
    return p.parse_args()


def main():
    args = parse_args()
    from e4t.builders import build_models
    from e4t.cli_common import checked_load
    from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
    from e4t.schedulers import SCHEDULER_MAPPING
    from e4t.utils import AttributeDict, WhitespaceTokenizer, load_weight_offsets
    from e4t.vae import VAEDecoder
    dev = torch.device("cuda:0")
    print(f"device: {dev}")
    unet, enc, text, _ = build_models(dev, args.unet_variant, seed=args.seed or 0)       # 49408-row token table; the pipeline adds the placeholder row
    with torch.device(dev):
        vae = VAEDecoder().requires_grad_(False)
    cfg = AttributeDict(placeholder_token="*s", domain_class_token="art", domain_embed_scale=0.1)
    tok, sched = None, SCHEDULER_MAPPING[args.scheduler_type].stable_diffusion("epsilon" if args.unet_variant == "sd14" else "v_prediction")
    if not args.random_init:
        d = args.pretrained_model_name_or_path
        with open(os.path.join(d, "config.json")) as f:
            config = AttributeDict(json.load(f))
        e4t = AttributeDict(config.pretrained_args) if config.pretrained_args is not None else config          # inference.py:56-57
        cfg = AttributeDict(placeholder_token=e4t.placeholder_token, domain_class_token=e4t.domain_class_token,
                            domain_embed_scale=float(e4t.domain_embed_scale))
        base = e4t.pretrained_model_name_or_path
        if not base or not os.path.isdir(base):
            raise SystemExit(f"{d}/config.json names the base model {base!r}: a local directory with unet.pt / vae.pt / text_encoder.pt / tokenizer/ is needed")
        # every load is strict: a misnamed key must not leave random weights behind (e4t/utils.py:119-124)
        checked_load(unet, os.path.join(base, "unet.pt"), lambda k: "wo" in k)
        checked_load(text, os.path.join(base, "text_encoder.pt"))
        sd = {k: v for k, v in torch.load(os.path.join(base, "vae.pt"), map_location="cpu").items() if k.startswith(("decoder.", "post_quant_conv."))}
        missing, unexpected = vae.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"{base}/vae.pt: missing {missing[:5]} unexpected {list(unexpected)[:5]}")
        if os.path.exists(os.path.join(d, "unet.pt")):
            checked_load(unet, os.path.join(d, "unet.pt"))
        else:
            load_weight_offsets(unet, os.path.join(d, "weight_offsets.pt"))
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(os.path.join(base, "tokenizer"))
        if os.path.exists(os.path.join(base, "scheduler", "scheduler_config.json")):
            sched = SCHEDULER_MAPPING[args.scheduler_type].from_pretrained(base, subfolder="scheduler")
    else:
        tok = WhitespaceTokenizer(base_size=49408, model_max_length=77)
    pipe = StableDiffusionE4TPipeline(vae=vae, text_encoder=text, tokenizer=tok, unet=unet, e4t_encoder=enc, scheduler=sched,
                                      safety_checker=None, feature_extractor=None, e4t_config=cfg, requires_safety_checker=False)
    if not args.random_init:
        d = args.pretrained_model_name_or_path
        checked_load(enc, os.path.join(d, "encoder.pt"))
        if os.path.exists(os.path.join(d, "text_encoder.pt")):                                                  # inference.py:92-101
            checked_load(text, os.path.join(d, "text_encoder.pt"))        # written by --train_text_encoder: already holds the placeholder row
    unet.requires_grad_(False)
    enc.requires_grad_(False)
    if args.enable_xformers_memory_efficient_attention:
        pipe.enable_xformers_memory_efficient_attention()
    print("loaded pipeline")
    if args.image_path_or_url:
        image = Image.open(args.image_path_or_url).convert("RGB").resize((512, 512))                           # e4t/utils.py load_image
    else:
        image = Image.fromarray((torch.rand(512, 512, 3) * 255).to(torch.uint8).numpy())
    generator = torch.Generator(device=dev).manual_seed(args.seed) if args.seed else None
    prompts = args.prompt.split("::")
    all_images = []
    for prompt in prompts:
        all_images.extend(pipe(prompt, num_inference_steps=args.num_inference_steps, guidance_scale=args.guidance_scale, generator=generator,
                               image=image, num_images_per_prompt=args.num_images_per_prompt, height=args.height, width=args.width).images)
    image_grid(all_images, len(prompts), args.num_images_per_prompt).save(args.output)
    print(f"DONE! See `{args.output}` for the results!")


if __name__ == "__main__":
    main()
