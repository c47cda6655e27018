"""E4T domain tuning on MI355X — the reference's tuning_e4t.py surface (:26-63) on the native trainer: the whole UNet, the
weight offsets and the E4T encoder head train (:139-147), one image expanded to the batch (:266), VAE latents computed once
(:268-269), gradient-norm clipping at 1.0 (:329-335).  Saves `unet.pt`, `encoder.pt`, `config.json` (:220-240).

    python tuning_e4t.py --pretrained_model_name_or_path <pretrain output dir> --train_image_path img.png --output_dir out
    python tuning_e4t.py --synthetic_data --train_batch_size 16 --max_train_steps 30 --output_dir out      # random weights

--pretrained_model_name_or_path is a directory written by pretrain_e4t.py: its config.json names the base Stable Diffusion
directory, the placeholder token, the class token and the prompt template (:97,249-253); its weight_offsets.pt / encoder.pt are
loaded strictly on top of the base weights.  The saved config carries `pretrained_args` so inference.py finds the base model.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "e4t-diffusion_amd"))

import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="E4T domain tuning (MI355X-native)")
    p.add_argument("--pretrained_model_name_or_path", type=str, default=None, help="directory with config.json / weight_offsets.pt / encoder.pt")
    p.add_argument("--prompt_template", type=str, default=None, help="If None, take the template from the pretrained args")
    p.add_argument("--reg_lambda", type=float, default=1e-4)
    p.add_argument("--domain_embed_scale", type=float, default=0.1)
    p.add_argument("--train_image_path", type=str, default=None)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--learning_rate", type=float, default=1.6e-5)
    p.add_argument("--scale_lr", action="store_true")
    p.add_argument("--max_train_steps", type=int, default=15)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--output_dir", type=str, default="e4t-tuned")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--mixed_precision", type=str, default="bf16", choices=["no", "bf16"])
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--train_text_encoder", action="store_true")
    p.add_argument("--unfreeze_clip_vision", action="store_true")
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--checkpointing_steps", type=int, default=10000)
    p.add_argument("--dataloader_num_workers", type=int, default=0)
    p.add_argument("--lr_scheduler", type=str, default="constant",
                   choices=["linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup"])
    p.add_argument("--lr_warmup_steps", type=int, default=0)
    p.add_argument("--use_8bit_adam", action="store_true")
    p.add_argument("--report_to", type=str, default=None)
    p.add_argument("--revision", type=str, default=None)
    p.add_argument("--logging_dir", type=str, default="logs")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--synthetic_data", action="store_true")
    p.add_argument("--unet_variant", type=str, default="sd14", choices=["sd14", "sd21"])
    a = p.parse_args()
    if a.use_8bit_adam:
        p.error("--use_8bit_adam (bitsandbytes) is CUDA-only; the fused fp32 AdamW kernel is used")
    if a.gradient_accumulation_steps < 1:
        p.error("--gradient_accumulation_steps must be >= 1")
    return a


def setup(args, dev):
    """tuning_e4t.py:96-147,240-265: read the pre-trained run's config.json, load the base Stable Diffusion weights it names plus
    that run's weight offsets and encoder, add the placeholder token, derive the class-token / empty-prompt conditioning and the
    prompt templates from the PRE-TRAINED arguments, build the trainer over UNet + E4T encoder (+ text encoder)."""
    from e4t import cli_common as cc
    from e4t.trainer import E4TTrainer
    from e4t.utils import AttributeDict, load_config_from_pretrained
    src = args.pretrained_model_name_or_path
    if src:
        pretrained_args = load_config_from_pretrained(src)                                         # :97
        base = pretrained_args.pretrained_model_name_or_path
        if base is None and not args.synthetic_data:
            raise SystemExit(f"{src}/config.json names no base model (pretrained_model_name_or_path)")
    elif args.synthetic_data:
        pretrained_args = AttributeDict(placeholder_token="*s", domain_class_token="art", prompt_template="a photo of {placeholder_token}",
                                        pretrained_model_name_or_path=None, clip_model_name_or_path="ViT-H-14::laion2b_s32b_b79k")
        base = None
    else:
        raise SystemExit("--pretrained_model_name_or_path <dir written by pretrain_e4t.py> is required (or --synthetic_data)")
    unet, enc, text, vae = cc.build_models(dev, base, args.unet_variant, seed=args.seed or 0, freeze_clip_vision=not args.unfreeze_clip_vision,
                                           e4t_dir=src)                                             # :99-118
    tokenizer = cc.load_tokenizer(base, allow_offline_standin=args.synthetic_data or base is None,
                                  vocab_size=text.get_input_embeddings().weight.shape[0], max_len=text.config["max_len"])
    placeholder_token_id = cc.add_placeholder_token(tokenizer, text, pretrained_args.placeholder_token)   # :120-127
    text.requires_grad_(bool(args.train_text_encoder))                                              # :130-132
    class_token_id, empty_ids = cc.conditioning_ids(tokenizer, pretrained_args.domain_class_token)  # :249-250,278-284
    if args.prompt_template is None:
        args.prompt_template = pretrained_args.prompt_template                                      # :252-253
    prompt_templates = cc.resolve_prompt_templates(args.prompt_template)
    if args.enable_xformers_memory_efficient_attention:
        unet.enable_xformers_memory_efficient_attention()
    ga = args.gradient_accumulation_steps
    lr = args.learning_rate * (args.train_batch_size * ga if args.scale_lr else 1)                  # :149-156 (one process)
    tr = E4TTrainer(unet, enc, text, vae, lr=lr, domain_embed_scale=args.domain_embed_scale, reg_lambda=args.reg_lambda,
                    prediction_type=getattr(pretrained_args, "prediction_type", None) or "epsilon", class_token_id=class_token_id,
                    empty_prompt_ids=empty_ids.to(dev), device=dev, tuning=True, max_grad_norm=args.max_grad_norm)
    tr.prepare(args.train_batch_size)
    print(f"Number of Trainable Parameters: {tr.flat.numel * 1.e-6:.2f} M")
    import random
    rng = random.Random(args.seed)

    def prompts(bsz):                                                                               # :286-296
        ids, idx = cc.tokenize_prompts(tokenizer, prompt_templates, pretrained_args.placeholder_token, placeholder_token_id, bsz, rng)
        return ids.to(dev), idx.to(dev)
    return dict(unet=unet, enc=enc, text=text, vae=vae, tokenizer=tokenizer, trainer=tr, lr=lr, prompts=prompts, pretrained_args=pretrained_args,
                placeholder_token_id=placeholder_token_id, class_token_id=class_token_id, empty_ids=empty_ids, prompt_templates=prompt_templates)


def main():
    args = parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from e4t.optimization import LRSchedule
    from e4t.utils import save_config, save_e4t_encoder
    st = setup(args, dev)
    unet, enc, text, tr, lr = st["unet"], st["enc"], st["text"], st["trainer"], st["lr"]
    if args.seed is not None:
        torch.manual_seed(args.seed)
    ga = args.gradient_accumulation_steps
    B, res = args.train_batch_size, args.resolution
    g = torch.Generator(device=dev).manual_seed(args.seed or 0)
    pil_image_to_save = None
    if args.synthetic_data and not args.train_image_path:
        image = torch.rand((1, 3, res, res), generator=g, device=dev) * 2 - 1
    elif args.train_image_path:
        # tuning_e4t.py:174-181: make_transforms(resolution, random_crop=True) on the one training image; decode on the host,
        # SmallestMaxSize(INTER_AREA) / crop / flip / normalise in the data-path kernel
        import random

        import numpy as np
        from PIL import Image
        from e4t import ops
        from e4t.data import make_transforms, pack_batch
        pil_image_to_save = Image.open(args.train_image_path).convert("RGB")
        rgb = np.ascontiguousarray(np.asarray(pil_image_to_save, dtype=np.uint8))
        plan = make_transforms(res, random_crop=True).plan(rgb.shape[0], rgb.shape[1], random.Random(args.seed))
        pool, table, _ = pack_batch([dict(image=rgb, plan=plan)], res)
        image = ops.backend().image_prep(pool.to(dev), table.to(dev), 1, res)
    else:
        raise SystemExit("give --train_image_path <file> or --synthetic_data")
    pixels = image.expand(B, -1, -1, -1).contiguous()                       # tuning_e4t.py:266
    latents = tr.encode_latents(pixels, torch.randn((B, 4, res // 8, res // 8), generator=g, device=dev))   # once, :268-269
    sched = LRSchedule(args.lr_scheduler, lr, args.lr_warmup_steps * ga, args.max_train_steps * ga)

    def save(d):                                                             # tuning_e4t.py:220-240
        os.makedirs(d, exist_ok=True)
        torch.save(unet.state_dict(), os.path.join(d, "unet.pt"))
        save_e4t_encoder(enc, d)
        if args.train_text_encoder:
            torch.save(text.state_dict(), os.path.join(d, "text_encoder.pt"))
        save_config(dict(vars(args), pretrained_args=dict(st["pretrained_args"])), d)
        if pil_image_to_save is not None:
            pil_image_to_save.save(os.path.join(d, "domain.png"))
        print(f"[*] Weights saved at {d}")

    # the reference counts iterations (micro-batches) here too: tuning_e4t.py:270,341-343
    t0 = time.perf_counter()
    for global_step in range(1, args.max_train_steps + 1):
        sync = global_step % ga == 0
        if (global_step - 1) % ga == 0:
            sched.apply(tr)
        ids, pidx = st["prompts"](B)
        loss, ld, lr_ = tr.train_step(pixels, ids, pidx, latents=latents, sync=sync, loss_scale=1.0 / ga)
        if sync:
            sched.step()
        torch.cuda.synchronize()
        print(f"step {global_step}: train/loss {float(loss):.5f} train/loss_diff {float(ld):.5f} train/loss_reg {float(lr_):.5f} train/lr {tr.lr:.3e} "
              f"{B * global_step / (time.perf_counter() - t0):.1f} img/s", flush=True)
        if global_step % args.checkpointing_steps == 0:
            save(os.path.join(args.output_dir, str(global_step)))
    save(os.path.join(args.output_dir, str(args.max_train_steps)))           # :346-348 saves <output_dir>/<max_train_steps>


if __name__ == "__main__":
    main()
