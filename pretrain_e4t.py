"""E4T pre-training on MI355X — same command-line surface as the reference's pretrain_e4t.py (:66-122), driving the
native modules (e4t-diffusion_amd/e4t) and trainer.  One process per GPU:

    python pretrain_e4t.py --synthetic_data --train_batch_size 16 --max_train_steps 100 --output_dir out
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 pretrain_e4t.py --synthetic_data ...
    accelerate launch pretrain_e4t.py ...        # LOCAL_RANK / WORLD_SIZE from the launcher are honoured

Without network access `--pretrained_model_name_or_path` is a LOCAL directory (e4t/cli_common.py: unet.pt / vae.pt /
text_encoder.pt state dicts with the diffusers / transformers key names, tokenizer/); without it the weights are randomly
initialised and an offline tokenizer stands in (--synthetic_data).  The set-up is the reference's: the placeholder token is
added to the tokenizer and the embedding table grown by it (:253-259), the class token id comes from --domain_class_token and
the E4T encoder pass is conditioned on tokenizer("") (:561-583), prompts are drawn from the reference's template lists.
Artefacts keep the reference's names: `{output_dir}/{step}/config.json`, `weight_offsets.pt`, `encoder.pt` (:515-528).
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "e4t-diffusion_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

def parse_args():
    p = argparse.ArgumentParser(description="E4T pre-training (MI355X-native)")
    p.add_argument("--pretrained_model_name_or_path", type=str, default=None)
    p.add_argument("--clip_model_name_or_path", type=str, default="ViT-H-14::laion2b_s32b_b79k")
    p.add_argument("--domain_class_token", type=str, default="art")
    p.add_argument("--domain_embed_scale", type=float, default=0.1)
    p.add_argument("--placeholder_token", type=str, default="*s")
    p.add_argument("--reg_lambda", type=float, default=0.01)
    p.add_argument("--prompt_template", type=str, default="a photo of {placeholder_token}",
                   help="{placeholder_token} is replaced by the placeholder token; 'normal' / 'face' / 'art' select the reference's template lists")
    p.add_argument("--unfreeze_clip_vision", action="store_true")
    p.add_argument("--train_image_dataset", type=str, default=None)
    p.add_argument("--webdataset", action="store_true")
    p.add_argument("--iterable_dataset", action="store_true")
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--learning_rate", type=float, default=1e-6)
    p.add_argument("--scale_lr", action="store_true")
    p.add_argument("--lr_scheduler", type=str, default="constant",
                   choices=["linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup"])
    p.add_argument("--lr_warmup_steps", type=int, default=0)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--max_train_steps", type=int, default=30000)
    p.add_argument("--dataloader_num_workers", type=int, default=0)
    p.add_argument("--output_dir", type=str, default="e4t-model")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--revision", type=str, default=None, help="accepted for command-line compatibility (no hub access here)")
    p.add_argument("--use_8bit_adam", action="store_true")
    p.add_argument("--mixed_precision", type=str, default="bf16", choices=["no", "fp16", "bf16"])
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--checkpointing_steps", type=int, default=10000)
    p.add_argument("--log_steps", type=int, default=1000, help="sample images at step 1 and every N steps; 0 disables sampling")
    p.add_argument("--save_sample_prompt", type=str, default="a photo of *s,a photo of *s in the style of monet")
    p.add_argument("--n_save_sample", type=int, default=4)
    p.add_argument("--save_guidance_scale", type=float, default=7.5)
    p.add_argument("--save_inference_steps", type=int, default=50)
    p.add_argument("--report_to", type=str, default="none")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--resume_from_checkpoint", type=str, default=None)
    p.add_argument("--prediction_type", type=str, default="epsilon", choices=["epsilon", "v_prediction"])
    # extensions
    p.add_argument("--synthetic_data", action="store_true", help="device-resident random images (benchmarking, CI); prompts still come from the templates")
    p.add_argument("--per_rank_seed", action="store_true", help="seed + rank for the step's random draws (the reference seeds every rank alike, :230-231)")
    p.add_argument("--unet_variant", type=str, default="sd14", choices=["sd14", "sd21"])
    args = p.parse_args()
    if args.use_8bit_adam:
        p.error("--use_8bit_adam (bitsandbytes) is CUDA-only and not part of the MI355X path; the fused fp32 AdamW kernel is used")
    if args.mixed_precision == "fp16":
        p.error("the native kernels compute in bf16 with fp32 accumulation; use --mixed_precision bf16")
    if args.gradient_accumulation_steps < 1:
        p.error("--gradient_accumulation_steps must be >= 1")
    env_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_rank != -1:
        args.local_rank = env_rank
    return args


def synthetic_batches(args, dev, rank, world, prompts):
    g = torch.Generator(device=dev)
    step = 0
    while True:
        g.manual_seed((args.seed or 0) * 100003 + step * world + rank)
        px = torch.rand((args.train_batch_size, 3, args.resolution, args.resolution), generator=g, device=dev) * 2 - 1
        ids, pidx = prompts(args.train_batch_size)
        yield px, ids, pidx
        step += 1


def image_batches(args, dev, rank, world, prompts):
    """real images (reference: E4TDataset + DataLoader, pretrain_e4t.py:147-180,284-291) through e4t.data: the host only
    decodes; SmallestMaxSize(INTER_AREA)/crop/flip/normalise run in one kernel per batch, prefetched under the step"""
    from e4t.data import DeviceLoader, E4TDataset, TarShardDataset, get_dataset_size
    if args.webdataset:
        n, nshards = get_dataset_size(args.train_image_dataset)
        print(f"Loading webdataset with {nshards} shards. (num_samples: {n})")
        ds = TarShardDataset(args.train_image_dataset, resolution=args.resolution)
    else:
        ds = E4TDataset(args.train_image_dataset, resolution=args.resolution)
    loader = DeviceLoader(ds, args.train_batch_size, shuffle=True, num_workers=args.dataloader_num_workers, device=dev,
                          rank=rank, world=world, seed=args.seed or 0)
    if not args.webdataset and len(loader) == 0:
        raise SystemExit(f"{len(ds)} images are fewer than one global batch")
    while True:
        for batch in loader:
            ids, pidx = prompts(args.train_batch_size)
            yield batch["pixel_values"], ids, pidx


def setup(args, dev, world=1, rank=0):
    """Everything between the argument parser and the loop (pretrain_e4t.py:233-259,274-278,354-361,561-583): models, tokenizer
    with the placeholder token, class-token / empty-prompt conditioning, prompt templates, learning rate, trainer."""
    from e4t import cli_common as cc
    from e4t.trainer import E4TTrainer
    base = args.pretrained_model_name_or_path
    e4t_dir = base if (base and os.path.exists(os.path.join(base, "weight_offsets.pt"))) else None      # :238-249
    # weights: a common seed on every rank (replicas must start identical: there is no DDP broadcast)
    unet, enc, text, vae = cc.build_models(dev, base, args.unet_variant, seed=args.seed or 0, freeze_clip_vision=not args.unfreeze_clip_vision,
                                           e4t_dir=e4t_dir)
    tokenizer = cc.load_tokenizer(base, allow_offline_standin=args.synthetic_data or base is None,
                                  vocab_size=text.get_input_embeddings().weight.shape[0], max_len=text.config["max_len"])
    placeholder_token_id = cc.add_placeholder_token(tokenizer, text, args.placeholder_token)              # :253-259
    class_token_id, empty_ids = cc.conditioning_ids(tokenizer, args.domain_class_token)                  # :561-569
    prompt_templates = cc.resolve_prompt_templates(args.prompt_template)                                  # :570-581
    if args.enable_xformers_memory_efficient_attention:
        unet.enable_xformers_memory_efficient_attention()        # selects the native flash-attention processor
    lr = args.learning_rate
    if args.scale_lr:                                                                                      # :354-361
        lr = args.learning_rate * args.gradient_accumulation_steps * args.train_batch_size * world
        print("Setting learning rate to {:.2e} = {} (accumulate_grad_batches) * {} (num_gpus) * {} (batchsize) * {:.2e} (base_lr)".format(
            lr, args.gradient_accumulation_steps, world, args.train_batch_size, args.learning_rate))
    tr = E4TTrainer(unet, enc, text, vae, lr=lr, domain_embed_scale=args.domain_embed_scale, reg_lambda=args.reg_lambda,
                    prediction_type=args.prediction_type, class_token_id=class_token_id, empty_prompt_ids=empty_ids.to(dev), device=dev)
    tr.prepare(args.train_batch_size)        # graph captures happen here, before any loader thread exists
    rng = random.Random((args.seed or 0) + (rank if args.per_rank_seed else 0)) if args.seed is not None else random.Random()

    def prompts(bsz):                                                                                      # :607-615
        ids, idx = cc.tokenize_prompts(tokenizer, prompt_templates, args.placeholder_token, placeholder_token_id, bsz, rng)
        return ids.to(dev), idx.to(dev)
    return dict(unet=unet, enc=enc, text=text, vae=vae, tokenizer=tokenizer, placeholder_token_id=placeholder_token_id,
                class_token_id=class_token_id, empty_ids=empty_ids, prompt_templates=prompt_templates, trainer=tr, lr=lr, prompts=prompts)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = max(args.local_rank, 0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
    from e4t.optimization import LRSchedule
    from e4t.utils import save_config, save_e4t_encoder, save_e4t_unet
    st = setup(args, dev, world, rank)
    unet, enc, text, vae, tr, lr = st["unet"], st["enc"], st["text"], st["vae"], st["trainer"], st["lr"]
    # the step's random draws (noise, timesteps, VAE sampling): the reference seeds every rank alike (set_seed, :230-231);
    # --per_rank_seed decorrelates them, no --seed leaves them unseeded
    if args.seed is not None:
        torch.manual_seed(args.seed + (rank if args.per_rank_seed else 0))
        random.seed(args.seed + (rank if args.per_rank_seed else 0))
    else:
        torch.seed()
    if args.synthetic_data:
        data = synthetic_batches(args, dev, rank, world, st["prompts"])
    elif args.train_image_dataset and not args.iterable_dataset:
        data = image_batches(args, dev, rank, world, st["prompts"])
    else:
        raise SystemExit("give --train_image_dataset <dir[::dir]>, --webdataset --train_image_dataset <shards{000..NNN}.tar>, or "
                         "--synthetic_data; HF-hub streaming (--iterable_dataset) needs network access")

    ga = args.gradient_accumulation_steps
    # reference :402-408: warm-up and horizon are given in micro-steps, and accelerate's scheduler wrapper advances the schedule
    # once per PROCESS on every optimiser step (AcceleratedScheduler.step, split_batches=False) — mirrored below
    sched = LRSchedule(args.lr_scheduler, lr, args.lr_warmup_steps * ga, args.max_train_steps * ga)
    first_step = 1
    if args.resume_from_checkpoint:                                                              # reference :536-558
        path = args.resume_from_checkpoint
        if path == "latest":
            dirs = [d for d in (os.listdir(args.output_dir) if os.path.isdir(args.output_dir) else []) if d.startswith("checkpoint-")]
            path = os.path.join(args.output_dir, max(dirs, key=lambda d: int(d.split("-")[1]))) if dirs else None
        found = path is not None and os.path.exists(os.path.join(path, "trainer_state.pt"))
        if world > 1:
            # load_state_dict() broadcasts (collectives): every rank must take the same branch.  Rank 0 decides; a rank that cannot see
            # what rank 0 sees (no shared filesystem, half-written directory) fails loudly instead of hanging the others.
            flag = torch.tensor([int(found)], device=dev)
            mine = int(flag)
            dist.broadcast(flag, src=0)
            found = bool(int(flag))
            agree = torch.tensor([int(mine == int(flag))], device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if not int(agree):
                raise SystemExit(f"--resume_from_checkpoint {args.resume_from_checkpoint}: the ranks disagree on whether '{path}' exists "
                                 "(rank 0 decides; give every rank the same view of the checkpoint directory)")
        if not found:
            print(f"Checkpoint '{args.resume_from_checkpoint}' does not exist. Starting a new training run.")
        else:
            print(f"Resuming from checkpoint {path}")
            tr.load_state_dict(torch.load(os.path.join(path, "trainer_state.pt"), map_location="cpu"))
            first_step = tr.step_count * ga + 1
            sched.step_count = tr.step_count * world

    def save(step, state=False):
        if rank != 0:
            return
        d = os.path.join(args.output_dir, str(step))
        save_config(vars(args), d)
        save_e4t_unet(unet, d)
        save_e4t_encoder(enc, d)
        if state:       # accelerator.save_state (:660-662): parameters + Adam moments + step, enough to resume bit for bit
            cd = os.path.join(args.output_dir, f"checkpoint-{step}")
            os.makedirs(cd, exist_ok=True)
            torch.save(tr.state_dict(), os.path.join(cd, "trainer_state.pt"))
            print(f"Saved state to {cd}")

    pipe_parts = {}

    @torch.no_grad()
    def sample(images, step):
        """qualitative logging (reference pretrain_e4t.py:452-513): for every prompt x a few training images, run the E4T
        pipeline and write input-<step>.png / sample-<step>.png grids under <output_dir>/samples"""
        from PIL import Image
        from e4t.pipeline_stable_diffusion_e4t import StableDiffusionE4TPipeline
        from e4t.schedulers import DDIMScheduler
        from e4t.vae import VAEDecoder
        from inference import image_grid
        if not pipe_parts:
            with torch.device(dev):
                dec = VAEDecoder().requires_grad_(False)
            f = os.path.join(args.pretrained_model_name_or_path or "", "vae.pt")
            if os.path.exists(f):       # the AutoencoderKL state dict: keep the decoder half
                sd = {k: v for k, v in torch.load(f, map_location="cpu").items() if k.startswith(("decoder.", "post_quant_conv."))}
                missing, unexpected = dec.load_state_dict(sd, strict=False)
                if missing or unexpected:
                    raise RuntimeError(f"{f}: missing {missing[:5]} unexpected {list(unexpected)[:5]}")
            pipe_parts.update(tok=st["tokenizer"], dec=dec, sched=DDIMScheduler.stable_diffusion(args.prediction_type))
        x = torch.clamp((images + 1.0) / 2.0, min=0.0, max=1.0)
        pils = [Image.fromarray((255.0 * xi.permute(1, 2, 0).cpu().numpy()).astype("uint8")) for xi in x]
        pils = random.sample(pils, min(len(pils), args.n_save_sample))
        was_training = unet.training
        pipe = StableDiffusionE4TPipeline(vae=pipe_parts["dec"], text_encoder=text, tokenizer=pipe_parts["tok"], unet=unet, e4t_encoder=enc,
                                          scheduler=pipe_parts["sched"], e4t_config=args, already_added_placeholder_token=True)
        prompts = args.save_sample_prompt.split(",")
        outs = [pipe(pr, guidance_scale=args.save_guidance_scale, num_inference_steps=args.save_inference_steps, image=im,
                     height=args.resolution, width=args.resolution).images[0] for pr in prompts for im in pils]
        d = os.path.join(args.output_dir, "samples")
        os.makedirs(d, exist_ok=True)
        image_grid(pils, rows=1, cols=len(pils)).save(os.path.join(d, f"input-{step}.png"))
        image_grid(outs, rows=len(prompts), cols=len(pils)).save(os.path.join(d, f"sample-{step}.png"))
        unet.train(was_training)

    # The reference counts ITERATIONS: global_step advances once per micro-batch (pretrain_e4t.py:656-657), so --max_train_steps,
    # --checkpointing_steps and --log_steps are in micro-batches; the optimiser steps every `ga`-th one (accelerator.accumulate).
    t_mark, step_mark = time.perf_counter(), first_step - 1
    # one batch of look-ahead (the loader prefetches two deep anyway): the trainer starts the frozen CLIP-ViT of the NEXT batch under
    # the current step's backward (E4TTrainer.prefetch; E4T_PREFETCH=0 switches it off)
    pending = next(data) if first_step <= args.max_train_steps else None
    for global_step in range(first_step, args.max_train_steps + 1):
        sync = global_step % ga == 0
        if (global_step - 1) % ga == 0:
            sched.apply(tr)
        batch = pending
        pending = next(data) if global_step < args.max_train_steps else None
        if pending is not None:
            tr.prefetch(pending[0])
        loss, ld, lr_ = tr.train_step(*batch, sync=sync, loss_scale=1.0 / ga)
        if sync:
            for _ in range(world):
                sched.step()
        if global_step % args.checkpointing_steps == 0:
            save(global_step, state=True)
        if args.log_steps and (global_step == 1 or global_step % args.log_steps == 0) and rank == 0:      # :664-668
            sample(batch[0], global_step)
        if global_step % 10 == 0 or global_step == 1:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if rank == 0:                                   # rate over the window since the previous report (step 1 includes warm-up)
                print(f"step {global_step}: train/loss {float(loss):.5f} train/loss_diff {float(ld):.5f} train/loss_reg {float(lr_):.5f} "
                      f"train/lr {tr.lr:.3e}  {args.train_batch_size * world * (global_step - step_mark) / (now - t_mark):.1f} img/s", flush=True)
            t_mark, step_mark = now, global_step
    if world > 1:
        dist.barrier()
    save(args.max_train_steps)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
