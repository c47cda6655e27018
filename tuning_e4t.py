"""E4T domain tuning on MI355X — the reference's tuning_e4t.py surface (:26-63) on the native trainer: the whole UNet, the
weight offsets and the E4T encoder head train (:139-147), one image expanded to the batch (:266), VAE latents computed once
(:268-269), gradient-norm clipping at 1.0 (:329-335).  Saves `unet.pt`, `encoder.pt`, `config.json` (:220-240).

    python tuning_e4t.py --synthetic_data --train_batch_size 16 --max_train_steps 30 --output_dir out
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
    p.add_argument("--prompt_template", type=str, default="a photo of {placeholder_token}")
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
    if a.train_text_encoder:
        p.error("--train_text_encoder: the CLIP text encoder is outside the native hot path (SURVEY.md §2 #8) and stays frozen")
    return a


def main():
    args = parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    from bench import build_models
    from e4t.optimization import LRSchedule
    from e4t.trainer import E4TTrainer
    from e4t.utils import load_weight_offsets, save_config, save_e4t_encoder
    unet, enc, text, vae = build_models(dev, args.unet_variant, seed=args.seed or 0)
    src = args.pretrained_model_name_or_path
    if src and os.path.isdir(src):
        if os.path.exists(os.path.join(src, "weight_offsets.pt")):
            load_weight_offsets(unet, os.path.join(src, "weight_offsets.pt"))
        if os.path.exists(os.path.join(src, "encoder.pt")):
            enc.load_state_dict(torch.load(os.path.join(src, "encoder.pt"), map_location="cpu"))
    if args.unfreeze_clip_vision:
        enc.clip_vision.requires_grad_(True)
    ga = args.gradient_accumulation_steps
    lr = args.learning_rate * (args.train_batch_size * ga if args.scale_lr else 1)                 # tuning_e4t.py:183-186
    tr = E4TTrainer(unet, enc, text, vae, lr=lr, domain_embed_scale=args.domain_embed_scale, reg_lambda=args.reg_lambda,
                    class_token_id=1125, device=dev, tuning=True, max_grad_norm=args.max_grad_norm)
    B, res = args.train_batch_size, args.resolution
    g = torch.Generator(device=dev).manual_seed(args.seed or 0)
    if args.synthetic_data:
        image = torch.rand((1, 3, res, res), generator=g, device=dev) * 2 - 1
    elif args.train_image_path:
        # tuning_e4t.py:174-181: make_transforms(resolution, random_crop=True) on the one training image; decode on the host,
        # SmallestMaxSize(INTER_AREA) / crop / flip / normalise in the data-path kernel
        import random

        import numpy as np
        from PIL import Image
        from e4t import ops
        from e4t.data import make_transforms, pack_batch
        rgb = np.ascontiguousarray(np.asarray(Image.open(args.train_image_path).convert("RGB"), dtype=np.uint8))
        plan = make_transforms(res, random_crop=True).plan(rgb.shape[0], rgb.shape[1], random.Random(args.seed))
        pool, table, _ = pack_batch([dict(image=rgb, plan=plan)], res)
        image = ops.backend().image_prep(pool.to(dev), table.to(dev), 1, res)
    else:
        raise SystemExit("give --train_image_path <file> or --synthetic_data")
    pixels = image.expand(B, -1, -1, -1).contiguous()                       # tuning_e4t.py:266
    latents = tr.encode_latents(pixels, torch.randn((B, 4, res // 8, res // 8), generator=g, device=dev))   # once, :268-269
    ids = torch.randint(1000, 40000, (1, 77), generator=g, device=dev).expand(B, -1).contiguous()
    pidx = torch.full((B,), 4, device=dev)
    sched = LRSchedule(args.lr_scheduler, lr, args.lr_warmup_steps * ga, args.max_train_steps * ga)

    def save(d):
        os.makedirs(d, exist_ok=True)
        torch.save(unet.state_dict(), os.path.join(d, "unet.pt"))
        save_e4t_encoder(enc, d)
        save_config(dict(vars(args), pretrained_args={}), d)

    t0 = time.perf_counter()
    for step in range(1, args.max_train_steps + 1):
        sched.apply(tr)
        for micro in range(ga):
            loss, ld, lr_ = tr.train_step(pixels, ids, pidx, latents=latents, sync=micro == ga - 1, loss_scale=1.0 / ga)
        sched.step()
        torch.cuda.synchronize()
        print(f"step {step}: train/loss {float(loss):.5f} loss_diff {float(ld):.5f} loss_reg {float(lr_):.5f} lr {tr.lr:.3e} "
              f"{B * ga * step / (time.perf_counter() - t0):.1f} img/s", flush=True)
        if step % args.checkpointing_steps == 0:
            save(os.path.join(args.output_dir, str(step)))
    save(args.output_dir)


if __name__ == "__main__":
    main()
