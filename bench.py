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
import fnmatch
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
# the newest round whose counter pass (tools/profile_round.sh) is committed under profiles/
PROFILE_ROUND = next((r for r in ("r06", "r05", "r04", "r03", "r02") if os.path.exists(os.path.join(ROOT, "profiles", r + "_pmc_traffic.csv"))), "r03")
HBM_PEAK = 8.0e12                                          # HBM3E spec (6.3e12 achievable), same guide


def build_models(dev, model, seed):
    """the benchmark's models: e4t.builders (the factory the CLIs use too), token table incl. the placeholder row"""
    from e4t.builders import build_models as _build
    return _build(dev, model, seed, vocab_size=49409)


def cpu_baseline_and_parity(model, threads, dev):
    """The CPU leg.  BASELINE config 1 — full-size UNet + ViT-H-14 encoder + text encoder + VAE encoder, B=1, fp32, one
    training step (VAE encode, both UNet passes, E4T encoder, loss, backward, AdamW) of the oracle (oracle/e4t_oracle.py, a
    port of the reference's algorithm) on this box's host cores: 1 warm-up + 1 timed step.

    The warm-up step is not wasted: the SAME seeded weights and inputs first go through the native HIP path on the GPU
    (B=1) and through a stock torch.autocast(bf16) run of the oracle, and the three are compared with the protocol of
    tests/parity_step.py -> the `parity` block of the JSON line (loss, 13 encoder maps, every weight-offset / head gradient)."""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import parity_step as ps                      # tests/: the oracle is the checker and the timed CPU baseline, nothing else
    torch.set_num_threads(threads)
    case = ps.cases()["full_sd14" if model == "sd14" else "full_sd21"]
    o = ps.build_oracle(case)
    n = ps.build_native(case, o, dev)
    d = ps.make_data(case)
    nat = ps.native_leg(case, n, d, dev)
    del n
    torch.cuda.empty_cache()
    rep, _ = ps.evaluate(case, o, d, nat, dev, verbose=False, strict=False)      # its CPU fp32 oracle run = the warm-up step
    # (strict=False so that the JSON line is still printed; main() turns n_bad != 0 into a non-zero exit code)
    torch.cuda.empty_cache()
    par = dict(case=case.name, rule=rep["rule"], n_quantities=rep["n_quantities"], n_bad=rep["n_bad"], bad=rep["bad"],
               kink_elements_aligned=rep["kink_elements_aligned"], kink_sign_disagreements=rep.get("kink_sign_disagreements"),
               **{"kink_elements_within_1e-2": rep["kink_elements_within_1e-2"]})
    for kind, key in (("losses", "loss_rel"), ("enc_maps", "enc_maps_rel"), ("grads", "grad_rel"), ("other", "latents_embed_ehat_rel"),
                      ("adamw", "adamw_step_rel")):
        if kind in rep:
            w, q = rep[kind]["worst"], rep[kind]["tightest"]
            par[key] = dict(worst=w["native"], worst_name=w["name"], stock_autocast_same_quantity=w["autocast"],
                            tightest_fraction_of_bound=q["used"], count=rep[kind]["count"])
    # timed step, including the optimiser (pretrain_e4t.py:652-654)
    import e4t_oracle as orc
    opt = torch.optim.AdamW(orc.trainable_parameters(o["unet"], o["enc"]), lr=1e-6)
    t0 = time.perf_counter()
    ps.oracle_leg(case, o, d, collect=False)
    opt.step()
    dt = time.perf_counter() - t0
    cpu = dict(value=1.0 / dt, unit="images/s", cores=threads, kind="port",
               sample=f"oracle/e4t_oracle.py, full {model} UNet + ViT-H-14 encoder + CLIP text + VAE encoder, B=1, fp32, "
                      f"1 warm-up + 1 timed training step incl. AdamW ({dt:.1f} s)")
    return cpu, par


def secondary_configs(dev, steps=4, warmup=2):
    """BASELINE.json configs[3] and configs[4], driver-witnessed (rank 0, one GPU, after the timed region of the headline run):
      C4  tuning_e4t.py step — every UNet weight + weight offsets + E4T head train, gradient-norm clip, ONE image expanded over
          B = 16, VAE encoded once outside the loop (tuning_e4t.py:266-338): ms/step, images/s, MFMA fraction on 3.53 TFLOP/image;
      C5  the SD-2.x UNet config (ctx 1024, dh 64, linear projections, v-prediction) pre-training step at 768 px (96 x 96 latents,
          T = 9216 self-attention), B = 1 (= BASELINE's per-GPU batch) and B = 4: MFMA fraction on 7.13 TFLOP/image (hot path).
    Their parity evidence is the -m gpu tests `tuning_real_width`, `full_sd21` (tests/test_configs_gpu.py)."""
    from e4t.trainer import E4TTrainer
    out = {}

    def timed(fn, n):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    gen = torch.Generator(device=dev).manual_seed(4321)
    empty_ids = torch.tensor([[49406] + [49407] * 76], device=dev)
    # ---- C4
    unet, enc, text, vae = build_models(dev, "sd14", seed=0)
    tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, reg_lambda=0.1, class_token_id=1125, empty_prompt_ids=empty_ids, device=dev, tuning=True, max_grad_norm=1.0)
    B = 16
    image = torch.rand((1, 3, 512, 512), generator=gen, device=dev) * 2 - 1
    with torch.no_grad():
        lat1 = tr.encode_latents(image, torch.randn((1, 4, 64, 64), generator=gen, device=dev))          # tuning_e4t.py:268-269: once
    px, lat = image.expand(B, -1, -1, -1).contiguous(), lat1.expand(B, -1, -1, -1).contiguous()
    ids = torch.randint(0, 49000, (1, 77), generator=gen, device=dev).expand(B, -1).contiguous()
    pidx = torch.full((B,), 5, device=dev)
    sec = timed(lambda: tr.train_step(px, ids, pidx, latents=lat), steps)
    out["C4_tuning_sd14_512px_b16"] = dict(ms_per_step=sec * 1e3, images_per_s=B / sec, trainable_parameters=tr.flat.numel,
                                            step_mfma_frac=B / sec * 3.53e12 / MFMA_PEAK, flop_per_image=3.53e12,
                                            parity_test="tests/test_configs_gpu.py::test_tuning_step_matches_oracle[tuning_real_width]")
    del tr, unet, enc, text, vae
    torch.cuda.empty_cache()
    # ---- README recipe (reference README.md:34-54): the headline step with --unfreeze_clip_vision — the 632 M ViT-H-14 parameters train too
    # (ViT backward on the kernels: +0.67 TFLOP/image over the frozen tower's forward), B = 16
    unet, enc, text, vae = build_models(dev, "sd14", seed=0)
    enc.clip_vision.requires_grad_(True)
    tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, empty_prompt_ids=empty_ids, device=dev)
    B = 16
    px = torch.rand((B, 3, 512, 512), generator=gen, device=dev) * 2 - 1
    ids = torch.randint(0, 49000, (B, 77), generator=gen, device=dev)
    pidx = torch.randint(1, 20, (B,), generator=gen, device=dev)
    sec = timed(lambda: tr.train_step(px, ids, pidx), steps)
    out["README_pretrain_sd14_unfreeze_clip_vision_b16"] = dict(
        ms_per_step=sec * 1e3, images_per_s=B / sec, trainable_parameters=tr.flat.numel, flop_per_image=4.53e12,
        step_mfma_frac_necessary=B / sec * 3.39e12 / MFMA_PEAK, step_mfma_frac_whole_step=B / sec * 4.53e12 / MFMA_PEAK,
        parity_test="tests/test_configs_gpu.py::test_unfrozen_vit_step_matches_oracle[unfrozen_vit_full]")
    del tr, unet, enc, text, vae
    torch.cuda.empty_cache()
    # ---- C5
    unet, enc, text, vae = build_models(dev, "sd21", seed=0)
    tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, prediction_type="v_prediction", class_token_id=1125, empty_prompt_ids=empty_ids, device=dev)
    for B in (1, 4):
        px = torch.rand((B, 3, 768, 768), generator=gen, device=dev) * 2 - 1
        ids = torch.randint(0, 49000, (B, 77), generator=gen, device=dev)
        pidx = torch.randint(1, 20, (B,), generator=gen, device=dev)
        tr.enable_step_graph(False)
        eager = timed(lambda: tr.train_step(px, ids, pidx), steps)
        # the step as ONE replayed HIP graph (E4TTrainer.enable_step_graph: at these batch sizes the eager step is bound by the
        # host's ~2250 launches, not by the GPU); warm-up = 1 eager step + the capture; bitwise equal to the eager step
        # (tests/test_model_gpu.py::test_step_graph_replay_equals_eager)
        graphed = tr.enable_step_graph(True)
        sec = timed(lambda: tr.train_step(px, ids, pidx), steps) if graphed else eager
        out[f"C5_pretrain_sd21_768px_b{B}"] = dict(ms_per_step=sec * 1e3, images_per_s=B / sec, step_graph=bool(graphed and tr._step_graphs),
                                                    ms_per_step_eager_launches=eager * 1e3,
                                                    step_mfma_frac_necessary=B / sec * HOT_FLOP_PER_IMAGE["sd21"] / MFMA_PEAK,
                                                    step_mfma_frac_whole_step=B / sec * STEP_FLOP_PER_IMAGE["sd21"] / MFMA_PEAK,
                                                    parity_test="tests/test_configs_gpu.py::test_sd2_config_step_matches_oracle[full_sd21]")
    del tr, unet, enc, text, vae
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] block (tuning step, SD-2.x @768)")
    ap.add_argument("--materialise-head-grad", action="store_true", help="A/B: write the E4T head's 845 MB stacked weight gradient (batched GEMM + plain AdamW) instead of the factored update")
    ap.add_argument("--head-allreduce", action="store_true", help="N > 1: all-reduce the E4T head's 845 MB stacked weight gradient instead of gathering its factors (A/B)")
    ap.add_argument("--step-graph", action="store_true", help="A/B: the whole step replayed from one HIP graph (no next-batch prefetch: the two exclude each other)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_comm = world == 1 and os.environ.get("E4T_FORCE_COMM") == "1"     # one GPU, the collective path on (1-rank RCCL group): a code-path check
    if force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_comm:
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
        import ctypes
        ctypes.CDLL(None).fflush(None)      # every rank: RCCL's start-up banner leaves the C-stdio buffer now, not at exit behind rank 0's JSON line
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from e4t import ops
    from e4t.trainer import E4TTrainer
    hip = ops.backend()     # raises when libe4t_hip.so is missing: no fallback
    unet, enc, text, vae = build_models(dev, args.model, seed=0)
    # tokenizer("") of CLIP: BOS + EOS padding (pretrain_e4t.py:565-583); class token "art" = one id of the table
    empty_ids = torch.tensor([[49406] + [49407] * 76], device=dev)
    tr = E4TTrainer(unet, enc, text, vae, lr=1e-6 * args.batch * world, class_token_id=1125, empty_prompt_ids=empty_ids, device=dev,
                    head_factor_exchange=not args.head_allreduce)
    tr.factored_head_update = not args.materialise_head_grad
    if args.step_graph:
        tr.prefetch_mode = "0"
        assert tr.enable_step_graph(True)
    tr_prefetch = tr.prefetch_mode

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
    # Next-batch prefetch (E4TTrainer.prefetch): every step is told which images come next — as a training loop that holds its
    # loader's look-ahead batch does — and starts the frozen CLIP-ViT (E4T_PREFETCH=vit+vae: and the VAE encoder) for THEM under its
    # own backward.  Each of the K timed steps therefore still runs one ViT (+ VAE) pass (for the batch after it; the first one
    # consumes what the last warm-up step started, the last one prefetches a batch that is never used): same work per step.
    for s in range(args.warmup):
        tr.prefetch(pool[(s + 1) % len(pool)][0])
        tr.train_step(*pool[s % len(pool)])
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        tr.prefetch(pool[(args.warmup + s + 1) % len(pool)][0])
        loss, _, _ = tr.train_step(*pool[(args.warmup + s) % len(pool)])
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    ips = B * world * args.steps / dt

    roof = roof_hbm = None
    if not args.no_kernel_roofline:
        # two extra, instrumented steps (outside the timed region): HIP events around every instrumented launch, recorded on the
        # stream the kernel is launched on.  EVERY rank runs them — a training step contains the gradient all-reduce, a collective
        # rank 0 must not enter alone — and rank 0 records and reports.
        if rank == 0:
            hip.prof = []
        tr.comm_timing = {} if (world > 1 or force_comm) else None      # per-region all-reduce enqueue times + exposed wait of this step (N > 1)
        # a steady-state step: it consumes what the last timed step prefetched for it and prefetches the batch after it
        nxt = (args.warmup + args.steps) % len(pool)
        tr.prefetch(pool[(nxt + 1) % len(pool)][0])
        tr.train_step(*pool[nxt])
        torch.cuda.synchronize()
        prof_steady, hip.prof = hip.prof, ([] if rank == 0 else None)
        # ... and one step with the prefetch off (frozen encoders inside their own step, nothing on the side stream during the backward):
        # the same kernels without the other stream's workgroups competing for CUs and L2 — the kernel-quality view of the same launches
        mode, tr.prefetch_mode = tr.prefetch_mode, "0"
        tr._pref.clear()
        tr.train_step(*batch(10_000))
        torch.cuda.synchronize()
        tr.prefetch_mode = mode
        prof_alone, hip.prof = hip.prof, prof_steady
    comm = tr.comm_report() if ((world > 1 or force_comm) and not args.no_kernel_roofline) else None
    if rank == 0 and not args.no_kernel_roofline:
        def aggregate(prof):
            agg = {}
            for key, fl, nb, e0, e1 in prof:
                a = agg.setdefault(key, [0.0, 0.0, 0.0, 0])
                a[0] += fl; a[1] += nb; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += 1
            return agg
        # `roofline` describes the kernels in the step that has the device to itself (prefetch off for that step: the frozen encoders inside
        # their own step, nothing on the side stream during the backward) — events bracket a launch on its stream, so in the steady-state
        # step of the timed region they also count the time a kernel's workgroups wait behind the other stream's (the side stream's ViT
        # GEMMs read 3x slower there than alone): that view is reported next to it (`in_timed_configuration`), each with the rocprofv3
        # summary of its own command under profiles/
        agg_timed, agg = aggregate(hip.prof), aggregate(prof_alone)
        hip.prof = None
        # op label = kernel symbol: the 64 / 128 / 160 tiles carry their LDS stage count (e4t_gemm_plan_t.stages), a template argument of the symbol
        names = {}
        for t_, (bm_, wgm_, wgn_) in {64: (64, 2, 2), 128: (128, 4, 2), 160: (128, 4, 1)}.items():
            for st_ in (2, 3, 4):
                names[f"conv{t_}s{st_}"] = f"gemm_dma_kernel<{bm_}, {t_}, {wgm_}, {wgn_}, 1, {st_}, false, 64>"
                names[f"gemm{t_}s{st_}"] = f"gemm_dma_kernel<{bm_}, {t_}, {wgm_}, {wgn_}, 0, {st_}, false, 64>"
        names.update({
                 # (round 6: the stride-1 convs of the 256 x 256 plan run gemm_pps_kernel — 40 of the 45 launches per step; stride-2 / upsampling
                 # convs stay on gemm_pp_kernel<1, false, false>; the 256 x 128 x 32 plan's stride-1 convs run conv_strip_kernel<2>)
                 "conv512": "gemm_pps_kernel<false>", "gemm512": "gemm_pp_kernel<0, false, false>", "gemm_tn": "gemm_tn_kernel",
                 "conv2320": "gemm_pq_kernel<1, 320, false>", "gemm2320": "gemm_pq_kernel<0, 320, false>",
                 "conv5256": "conv_strip_kernel<2>", "gemm5256": "gemm_dma_kernel<256, 128, 4, 2, 0, 3, false, 32>",
                 "gn_fwd_colstats": "gn_stats_cols_kernel + gn_apply_kernel<true, *> (small maps: gn_slab_fwd_kernel<*>)",
                 "gn_fwd_2pass": "gn_stats_kernel<*> + gn_apply_kernel<true, *> (small maps: gn_slab_fwd_kernel<*>)",
                 "gn_bwd": "gn_bwd_stats_kernel<*> + gn_bwd_apply_kernel<*>",      # (small maps: gn_slab_bwd_kernel<*>)
                 "ln_fwd": "ln_fwd_kernel<*>", "ln_bwd": "ln_bwd_kernel<*>",
                 "geglu_fwd": "geglu_fwd_kernel", "geglu_bwd": "geglu_bwd_kernel", "adamw": "adamw_kernel"})
        # HBM bytes per launch of each kernel symbol from the TCC counters: collected offline with tools/profile_round.sh on this
        # very command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 on gfx950 as
        # MI355X_MICROARCH.md prescribes; check: adamw_kernel = 28 B x parameters) -> profiles/rNN_pmc_traffic.csv, whose header
        # line names the commit it was collected at; profiles/rNN_roofline_per_shape.csv breaks it down per shape.
        traffic_tab, traffic_src = {}, None
        csv_path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_prefetch_off_pmc_traffic.csv")
        if not os.path.exists(csv_path):
            csv_path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_pmc_traffic.csv")
        if os.path.exists(csv_path) and args.model == "sd14" and args.batch == 16:
            lines = open(csv_path).read().splitlines()
            traffic_src = "profiles/" + os.path.basename(csv_path) + (" " + lines[0].lstrip("# ") if lines and lines[0].startswith("#") else "")
            for line in lines:
                if line.startswith("#") or line.startswith("kernel,"):
                    continue
                parts = line.rsplit(",", 4)
                traffic_tab[parts[0].strip('"')] = (int(parts[1]), float(parts[4]))
        RIDGE = MFMA_PEAK / HBM_PEAK                 # 312 FLOP/B: below it a kernel cannot be MFMA-bound

        def op_traffic(key):
            """HBM bytes per launch of the op: one symbol -> its row; an op that always launches one kernel of each of several
            templated symbols ("a<*> + b<*>") -> sum of their bytes over the launches of the first.  The two GroupNorm forward
            paths share gn_apply_kernel<true, *>, so the symbol table cannot split them (per-shape table does): None."""
            name = names.get(key, key)
            if key in ("gn_fwd_colstats", "gn_fwd_2pass"):
                return None
            parts = [q.strip() for q in name.split("+")]
            rows = [[v for k, v in traffic_tab.items() if (fnmatch.fnmatchcase(k, q) if "*" in q else k == q)] for q in parts]
            if not all(rows):
                return None
            return sum(n * b for r in rows for n, b in r) / sum(n for n, _ in rows[0])

        def entry(key):
            fl, nb, sec, n = agg[key]
            intensity = fl / nb if nb else 0.0
            d = dict(kernel=names.get(key, key), launches_per_step=n, avg_launch_ms=sec / n * 1e3, ms_per_step=sec * 1e3,
                     algorithmic_flop_per_launch=fl / n, algorithmic_bytes_per_launch=nb / n, intensity_flop_per_byte=intensity)
            if intensity >= RIDGE:
                d.update(bound="mfma", achieved=fl / sec / 1e12, peak=MFMA_PEAK / 1e12, unit="TFLOP/s", frac=fl / sec / MFMA_PEAK)
            else:
                d.update(bound="hbm", achieved=nb / sec / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=nb / sec / HBM_PEAK)
                if fl:
                    d["mfma_frac"] = fl / sec / MFMA_PEAK
            d["traffic"] = op_traffic(key)
            if d["traffic"]:
                d["traffic_over_algorithmic"] = d["traffic"] / (nb / n)
            return d

        dom = max(agg, key=lambda k: agg[k][2])
        roof = entry(dom)
        roof["traffic_source"] = traffic_src
        roof["configuration"] = ("one instrumented step with the next-batch prefetch off for that step (frozen ViT / VAE inside their own step, ViT on its side "
                                 "stream under the UNet encoder pass as in round 3), events on the launch stream; same command as profiles/" + PROFILE_ROUND +
                                 "_prefetch_off_step_kernel_stats.csv (E4T_PREFETCH=0 python bench.py)")
        if dom in agg_timed and agg_timed[dom][2] > 0:
            fl, nb, sec, n = agg_timed[dom]
            roof["in_timed_configuration"] = dict(
                configuration="the same kernel in a steady-state step of the timed region (next-batch prefetch " + tr_prefetch + ": the frozen ViT / VAE of batch i+1 "
                              "run on the side stream under this step and compete for CUs); same command as profiles/" + PROFILE_ROUND + "_step_kernel_stats.csv",
                launches_per_step=n, avg_launch_ms=sec / n * 1e3, ms_per_step=sec * 1e3,
                achieved=(fl / sec / 1e12 if roof["bound"] == "mfma" else nb / sec / 1e9),
                frac=(fl / sec / MFMA_PEAK if roof["bound"] == "mfma" else nb / sec / HBM_PEAK))
        roof["per_kernel"] = {k: dict(tflops=v[0] / v[2] / 1e12, gbps=v[1] / v[2] / 1e9, ms_per_step=v[2] * 1e3, launches=v[3],
                                      intensity=(v[0] / v[1] if v[1] else 0.0)) for k, v in sorted(agg.items())}
        roof["per_kernel_in_timed_configuration"] = {k: dict(tflops=v[0] / v[2] / 1e12, gbps=v[1] / v[2] / 1e9, ms_per_step=v[2] * 1e3, launches=v[3])
                                                     for k, v in sorted(agg_timed.items())}
        roof["step_mfma_frac_necessary"] = ips * HOT_FLOP_PER_IMAGE[args.model] / (world * MFMA_PEAK)
        roof["step_mfma_frac_whole_step"] = ips * STEP_FLOP_PER_IMAGE[args.model] / (world * MFMA_PEAK)
        pure_hbm = [k for k in agg if agg[k][0] == 0.0]
        if pure_hbm:
            roof_hbm = entry(max(pure_hbm, key=lambda k: agg[k][2]))
    if world > 1:
        dist.barrier()

    cpu = parity = secondary = None
    if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_secondary):
        del tr, unet, enc, text, vae
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_secondary and args.model == "sd14":
        secondary = secondary_configs(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or min(os.cpu_count() or 8, 128)
        cpu, parity = cpu_baseline_and_parity(args.model, threads, dev)

    if rank == 0:
        # The printed line is SHORT (the driver's record keeps the known keys and the last ~2 KB of stdout: the 16 KB line of rounds 2-4 lost
        # its parity / secondary / per-kernel blocks there).  Everything else goes to gpurun_out/bench_details.json (+ stderr).
        details = dict(roofline=roof, roofline_hbm=roof_hbm, parity=parity, secondary=secondary, comm=comm)
        r3 = lambda x: None if x is None else float(f"{x:.4g}")
        short_roof = short_hbm = None
        if roof is not None:
            keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "launches_per_step", "avg_launch_ms",
                    "ms_per_step", "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch", "step_mfma_frac_necessary", "step_mfma_frac_whole_step")
            short_roof = {k: (r3(v) if isinstance(v, float) else v) for k, v in roof.items() if k in keep}
            short_roof["configuration"] = "instrumented step with the next-batch prefetch off (= profiles/" + PROFILE_ROUND + "_prefetch_off_step_kernel_stats.csv)"
            itc = roof.get("in_timed_configuration")
            if itc:
                short_roof["in_timed_configuration"] = dict(frac=r3(itc["frac"]), avg_launch_ms=r3(itc["avg_launch_ms"]), achieved=r3(itc["achieved"]))
            # ms per step of every op family in that step (the per-kernel table with TF/s, GB/s and launch counts: bench_details.json)
            short_roof["ms_per_step_by_op"] = {k: round(v["ms_per_step"], 2) for k, v in sorted(roof["per_kernel"].items(), key=lambda kv: -kv[1]["ms_per_step"])}
        if roof_hbm is not None:
            short_hbm = {k: (r3(v) if isinstance(v, float) else v) for k, v in roof_hbm.items()
                         if k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "ms_per_step")}
        short_par = None
        if parity is not None:
            g = parity.get("grad_rel") or {}
            short_par = dict(case=parity["case"], n_bad=parity["n_bad"], n_quantities=parity["n_quantities"],
                             kink_band=(parity["kink_elements_aligned"] or {}).get("band"), kink_elements_aligned=(parity["kink_elements_aligned"] or {}).get("native"),
                             worst_grad=r3(g.get("worst")), worst_grad_stock_autocast=r3(g.get("stock_autocast_same_quantity")),
                             tightest_fraction_of_bound=r3(g.get("tightest_fraction_of_bound")), bad=[b["name"] for b in parity["bad"][:4]])
        sec_ms = sec_ips = None
        if secondary is not None:
            names = {"C4_tuning_sd14_512px_b16": "C4", "README_pretrain_sd14_unfreeze_clip_vision_b16": "README", "C5_pretrain_sd21_768px_b1": "C5_b1",
                     "C5_pretrain_sd21_768px_b4": "C5_b4"}
            sec_ms = {names.get(k, k): round(v["ms_per_step"], 2) for k, v in secondary.items()}
            sec_ips = {names.get(k, k): round(v["images_per_s"], 1) for k, v in secondary.items()}
        out = dict(metric="E4T pretrain images/sec @512px bf16" if args.model == "sd14" else "E4T pretrain images/sec @768px bf16",
                   value=ips, unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload=("SD-1.4 UNet + ViT-H-14 E4T encoder pretrain step, 512px" if args.model == "sd14"
                                         else "SD-2.x UNet (ctx 1024, linear proj) + ViT-H-14 E4T encoder pretrain step, 768px"),
                               per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}", trainable="weight offsets + E4T head (ViT frozen)",
                               next_batch_prefetch=tr_prefetch, last_loss=float(loss)),
                   roofline=short_roof, roofline_hbm=short_hbm, cpu_baseline=cpu, parity=short_par,
                   secondary_ms_per_step=sec_ms, secondary_images_per_s=sec_ips, comm=comm, details="gpurun_out/bench_details.json")
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w") as fh:
                json.dump(dict(line=out, **details), fh, indent=1)
        except OSError:
            out["details"] = None
        print(json.dumps(details), file=sys.stderr, flush=True)
    if world > 1 or force_comm:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; on a pipe that buffer is only flushed at exit, i.e. AFTER a line Python flushed.
        # Push it out first so that the JSON line is the last line of rank 0's stdout.
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if rank == 0 and parity is not None and parity["n_bad"]:
        sys.exit(3)                                  # the run's own parity block failed: not a valid measurement


if __name__ == "__main__":
    main()
