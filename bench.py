"""E4T pre-training throughput on MI355X (BASELINE.json metric) — driver contract in the task description.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full reference training step (pretrain_e4t.py:595-654) on a synthetic batch: VAE encode,
noise, UNet encoder pass, E4T encoder (ViT-H-14 + head), embed injection, text encoder, UNet full pass, loss,
backward, gradient all-reduce (N > 1), fused AdamW.  Workload at N=1 = BASELINE.json configs[1]: SD-1.4 UNet
config + ViT-H-14 E4T encoder, 512 px, bf16, per-GPU batch 16, random-init weights, synthetic images
("data": "synthetic").  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "e4t-diffusion_amd"), ROOT]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# necessary hot-path FLOP per image, SURVEY.md §8d / BASELINE.md §2 (UNet enc+full fwd, necessary bwd, ViT-H fwd, head)
HOT_FLOP_PER_IMAGE = {"sd14": 2.72e12, "sd21": 7.13e12}
STEP_FLOP_PER_IMAGE = {"sd14": 3.86e12, "sd21": 9.83e12}   # incl. frozen VAE encode + text encoder
MFMA_PEAK = 2.5e15                                         # bf16 dense, /opt/skills/guides/MI355X_MICROARCH.md


def build_models(dev, model, seed):
    from e4t.encoder import E4TEncoder
    from e4t.frozen import CLIP_TEXT_H, CLIP_TEXT_L
    from e4t.text import CLIPTextModel          # CLIP text encoder on the HIP kernels
    from e4t.vae import VAEEncoder
    from e4t.models.unet_2d_condition import UNet2DConditionModel
    torch.manual_seed(seed)
    base = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                norm_num_groups=32, norm_eps=1e-5)
    if model == "sd14":
        ucfg = dict(base, cross_attention_dim=768, attention_head_dim=8)
        tcfg, wdim = CLIP_TEXT_L, 768
    else:
        ucfg = dict(base, sample_size=96, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20), use_linear_projection=True)
        tcfg, wdim = CLIP_TEXT_H, 1024
    with torch.device(dev):
        unet = UNet2DConditionModel(**ucfg)
        enc = E4TEncoder(word_embedding_dim=wdim, block_out_channels=ucfg["block_out_channels"], arch="ViT-H-14")
        text = CLIPTextModel(**tcfg).requires_grad_(False)      # fp32 master weights; bf16 compute copies are made once
        vae = VAEEncoder().requires_grad_(False)          # fp32 masters; the kernel path keeps its own bf16 copies
    return unet, enc, text, vae


def cpu_baseline(model, threads):
    """The CPU oracle (a port of the reference's algorithm, oracle/e4t_oracle.py) timed on this box's host cores:
    BASELINE config 1 — full-size UNet + ViT-H-14 encoder, B=1, fp32, ONE training step (fwd, bwd, AdamW)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import e4t_oracle as orc
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    cfg = orc.SD14_UNET_CONFIG if model == "sd14" else orc.SD21_UNET_CONFIG
    wdim = cfg["cross_attention_dim"]
    unet = orc.UNet2DConditionModel(**cfg)
    enc = orc.E4TEncoder(word_embedding_dim=wdim)
    tcfg = orc.CLIP_TEXT_L if model == "sd14" else orc.CLIP_TEXT_H
    text = orc.CLIPTextModel(**tcfg).requires_grad_(False)
    vae = orc.VAEEncoder().requires_grad_(False)
    opt = torch.optim.AdamW(orc.trainable_parameters(unet, enc), lr=1e-6)
    acp = orc.ddpm_alphas_cumprod()
    res = 512 if model == "sd14" else 768
    g = torch.Generator().manual_seed(0)
    px = torch.rand(1, 3, res, res, generator=g) * 2 - 1
    ids = torch.randint(0, 49000, (1, 77), generator=g)
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([1125]))[0]
        ctx0 = text(input_ids=torch.zeros(1, 77, dtype=torch.long))
    t0 = time.perf_counter()
    with torch.no_grad():
        lat = vae.encode_sample(px, torch.randn(1, 4, res // 8, res // 8, generator=g))
        emb = text.get_input_embeddings()(ids)
    noise = torch.randn(lat.shape, generator=g)
    t = torch.randint(0, 1000, (1,), generator=g)
    loss, _, _, _ = orc.e4t_losses(unet, enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds), px, lat, noise, t, emb, [5], ctx0,
                                   class_embed, acp)
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="images/s", cores=threads, kind="port",
                sample=f"oracle/e4t_oracle.py, full {model} UNet + ViT-H-14 encoder, B=1, fp32, 1 untimed-warmup-free training step ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--model", default="sd14", choices=["sd14", "sd21"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-kernel-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from e4t import ops
    from e4t.trainer import E4TTrainer
    hip = ops.backend()     # raises when libe4t_hip.so is missing: no fallback
    unet, enc, text, vae = build_models(dev, args.model, seed=0)
    tr = E4TTrainer(unet, enc, text, vae, lr=1e-6 * args.batch * world, class_token_id=1125, device=dev)

    B = args.batch
    res = 512 if args.model == "sd14" else 768
    gen = torch.Generator(device=dev)

    def batch(step):
        gen.manual_seed(1234 + step * world + rank)
        px = torch.rand((B, 3, res, res), generator=gen, device=dev) * 2 - 1     # "WikiArt-shaped" after the reference's transforms
        ids = torch.randint(0, 49000, (B, 77), generator=gen, device=dev)
        pidx = torch.randint(1, 20, (B,), generator=gen, device=dev)
        return px, ids, pidx

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # synthetic input batches are generated up front and sit in HBM when the timed region starts (a pool of distinct
    # batches, cycled): the timed region is the training step, not torch's RNG
    pool = [batch(s) for s in range(min(4, args.warmup + args.steps))]
    for s in range(args.warmup):
        tr.train_step(*pool[s % len(pool)])
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        loss, _, _ = tr.train_step(*pool[(args.warmup + s) % len(pool)])
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    ips = B * world * args.steps / dt

    roof = None
    if not args.no_kernel_roofline:
        # one extra, instrumented step (outside the timed region): HIP events around every MFMA-kernel launch.  EVERY rank
        # runs it — a training step contains the gradient all-reduce, a collective rank 0 must not enter alone — and rank 0
        # records and reports.
        if rank == 0:
            hip.prof = []
        tr.overlap_vision = False      # per-kernel durations must not include a concurrently running side stream
        tr.train_step(*batch(10_000))
        torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_roofline:
        agg = {}
        for key, fl, e0, e1 in hip.prof:
            a = agg.setdefault(key, [0.0, 0.0, 0])
            a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        hip.prof = None
        dom = max(agg, key=lambda k: agg[k][1])
        fl, sec, n = agg[dom]
        names = {"conv128": "gemm_dma_kernel<128,128,4,2,conv,2> (implicit-GEMM 3x3 conv, 8 waves)",
                 "conv160": "gemm_dma_kernel<128,160,4,1,conv,2> (implicit-GEMM 3x3 conv)",
                 "conv64": "gemm_dma_kernel<64,64,2,2,conv,2> (implicit-GEMM 3x3 conv)",
                 "gemm128": "gemm_dma_kernel<128,128,4,2,dense,2>", "gemm160": "gemm_dma_kernel<128,160,4,1,dense,2>",
                 "gemm64": "gemm_dma_kernel<64,64,2,2,dense,2>",
                 "conv512": "gemm_pp_kernel<conv> (256x256 ping-pong implicit-GEMM 3x3 conv)", "gemm512": "gemm_pp_kernel<dense> (256x256 ping-pong)",
                 "gemm_tn": "gemm_tn_kernel (weight gradients, contraction over rows)"}
        # HBM bytes per launch of that kernel symbol from the TCC counters: collected offline with tools/pmc_traffic.sh on this
        # very command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 on gfx950 as
        # MI355X_MICROARCH.md prescribes; check: adamw_kernel comes out at 10.49 GB = 28 B x 374.5 M parameters)
        traffic = None
        sym = {"conv128": "gemm_dma_kernel<128, 128, 4, 2, 1, 2>", "conv160": "gemm_dma_kernel<128, 160, 4, 1, 1, 2>",
               "gemm128": "gemm_dma_kernel<128, 128, 4, 2, 0, 2>", "gemm160": "gemm_dma_kernel<128, 160, 4, 1, 0, 2>",
               "gemm64": "gemm_dma_kernel<64, 64, 2, 2, 0, 2>", "conv512": "gemm_pp_kernel<1>", "gemm512": "gemm_pp_kernel<0>",
               "gemm_tn": "gemm_tn_kernel"}.get(dom)
        csv_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.csv")
        if sym and os.path.exists(csv_path) and args.model == "sd14" and args.batch == 16:
            for line in open(csv_path).read().splitlines()[1:]:
                parts = line.rsplit(",", 4)
                if parts[0] == sym:
                    traffic = float(parts[4])
        roof = dict(bound="mfma", achieved=fl / sec / 1e12, peak=MFMA_PEAK / 1e12, unit="TFLOP/s", frac=fl / sec / MFMA_PEAK, traffic=traffic,
                    kernel=names.get(dom, dom), launches_per_step=n, avg_launch_ms=sec / n * 1e3,
                    per_kernel={k: dict(tflops=v[0] / v[1] / 1e12, ms_per_step=v[1] * 1e3, launches=v[2]) for k, v in sorted(agg.items())},
                    step_mfma_frac_necessary=ips * HOT_FLOP_PER_IMAGE[args.model] / (world * MFMA_PEAK),
                    step_mfma_frac_whole_step=ips * STEP_FLOP_PER_IMAGE[args.model] / (world * MFMA_PEAK))
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or min(os.cpu_count() or 8, 128)
        del tr, unet, enc, text, vae
        torch.cuda.empty_cache()
        cpu = cpu_baseline(args.model, threads)

    if rank == 0:
        out = dict(metric="E4T pretrain images/sec @512px bf16" if args.model == "sd14" else "E4T pretrain images/sec @768px bf16",
                   value=ips, unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload=("SD-1.4 UNet + ViT-H-14 E4T encoder pretrain step, 512px" if args.model == "sd14"
                                         else "SD-2.x UNet (ctx 1024, linear proj) + ViT-H-14 E4T encoder pretrain step, 768px"),
                               per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}", trainable="weight offsets + E4T head (ViT frozen)",
                               frozen_on_stock_torch="none (CLIP text encoder and VAE encoder run on the HIP kernels; only embedding lookups / loss glue are torch ops)", last_loss=float(loss)),
                   roofline=roof, cpu_baseline=cpu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
