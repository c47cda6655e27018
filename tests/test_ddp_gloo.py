"""CPU, world_size 2 over gloo: the data-parallel path of the trainer (one process per rank, all-reduce of the flat
gradient buffer, 1/N folded into AdamW) gives every rank the same parameters, equal to a single-process step that
accumulates both ranks' batches."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build():
    from test_train_step_host_logic import build
    from e4t import ops
    from emu_backend import EmuBackend
    ops.set_backend(EmuBackend(round_bf16=False))
    ops.ACT = torch.float32
    return build()


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    B = 1
    return dict(pixels=torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, latents=torch.randn(B, 4, 16, 16, generator=g) * 0.18215,
                noise=torch.randn(B, 4, 16, 16, generator=g), t=torch.randint(0, 1000, (B,), generator=g), ids=torch.randint(1, 99, (B, 9), generator=g),
                pidx=torch.tensor([3]))


def _perturb(models, rank):
    """make this rank's weights differ from rank 0's (a per-rank seed / a different checkpoint read): trainable AND frozen ones"""
    g = torch.Generator().manual_seed(900 + rank)
    with torch.no_grad():
        for m in models:
            for p in m.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 1e-2 * rank)


def _worker(rank, world, port, out_dir, unfreeze):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), E4T_REPLICA_CHECK_EVERY="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from e4t.trainer import E4TTrainer
    _, _, n_unet, n_enc, text = _build()
    if unfreeze == "vit":
        n_enc.clip_vision.requires_grad_(True)
    if unfreeze == "text":
        text.requires_grad_(True)
    if unfreeze in ("seeds", "text"):
        _perturb((n_unet, n_enc, text), rank)          # rank 1 starts from different weights: the start-up broadcast must repair it
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=torch.device("cpu"),
                    **_MODE_KW[unfreeze])
    assert tr.world == world
    if unfreeze == "accum":          # a micro-batch that only accumulates, then the synchronising one
        b = _batch(rank + 10)
        tr.train_step(b["pixels"], b["ids"], b["pidx"], noise=b["noise"], timesteps=b["t"], latents=b["latents"], sync=False)
    b = _batch(rank)
    tr.train_step(b["pixels"], b["ids"], b["pidx"], noise=b["noise"], timesteps=b["t"], latents=b["latents"])
    torch.save(tr.flat.data.clone(), os.path.join(out_dir, f"rank{rank}.pt"))
    torch.save(int(getattr(tr, "_factor_bytes", 0)), os.path.join(out_dir, f"factor_bytes{rank}.pt"))
    torch.save(tr.deferred_region, os.path.join(out_dir, f"deferred{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


_MODE_KW = {"": {}, "vit": {}, "tuning": dict(tuning=True, max_grad_norm=1.0), "seeds": {}, "text": dict(tuning=True, max_grad_norm=1.0),
            "allreduce_head": dict(head_factor_exchange=False), "accum": {}}
# the head's stacked weight gradient crosses the ranks as gathered factors (trainer._exchange_head_factors): everywhere except when it
# is switched off and when the stack already holds a micro-batch's local sum
_FACTORS = {"": True, "vit": True, "tuning": True, "seeds": True, "text": True, "allreduce_head": False, "accum": False}


@pytest.mark.parametrize("unfreeze", ["", "vit", "tuning", "seeds", "text", "allreduce_head", "accum"],
                         ids=["vit_frozen", "vit_trainable", "tuning", "different_seeds", "text_trainable", "head_stack_all_reduced", "micro_batches"])
def test_two_rank_step_matches_accumulated_single_process(tmp_path, unfreeze):
    """vit_trainable: no hook announces the head region during the backward, the post-backward sweep must reduce it.
    tuning: the whole UNet is in the U / D regions and the gradient-norm clip runs on the AVERAGED gradient (tuning_e4t.py:329-335
    under accelerate's DDP).
    different_seeds: rank 1 starts from perturbed weights (trainable and frozen); the trainer's start-up broadcast (the DDP
    constructor's, pretrain_e4t.py:410-412) makes it rank 0's replica — the ranks end identical AND equal to the single-process step
    from rank 0's weights; the per-step replica checksum (E4T_REPLICA_CHECK_EVERY=1) passes.
    text_trainable: tuning_e4t.py --train_text_encoder.  The token-embedding gradient is complete only at the END of the backward
    (region T, reduced by the sweep): reducing it with the E4T head's region from the head's hook dropped the late part.
    head_stack_all_reduced / micro_batches: the head's stacked weight gradient (region W) goes through the all-reduce — because the
    factor exchange is switched off, or because the stack already holds the first micro-batch's LOCAL sum; in every other mode the
    head's backward gathers both ranks' factors and writes the global sum itself (no all-reduce of W)."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), unfreeze), nprocs=world, join=True)
    p0, p1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(p0, p1), "ranks diverged after the all-reduced step"
    assert all((torch.load(tmp_path / f"factor_bytes{r}.pt") > 0) == _FACTORS[unfreeze] for r in range(world))
    # region D's all-reduce starts when the backward ends: AdamW of everything else runs under it — unless a gradient clip needs the whole norm first
    assert all(torch.load(tmp_path / f"deferred{r}.pt") == (None if "max_grad_norm" in _MODE_KW[unfreeze] else "D") for r in range(world))
    # single process: accumulate both batches' gradients, average inside AdamW
    _setup_paths()
    from e4t import ops
    old_b, old_act, old_threads = ops._backend, ops.ACT, torch.get_num_threads()
    try:
        from e4t.trainer import E4TTrainer
        torch.set_num_threads(2)       # the rank legs' setting: the same BLAS partitioning on both sides of the comparison
        _, _, n_unet, n_enc, text = _build()
        if unfreeze == "vit":
            n_enc.clip_vision.requires_grad_(True)
        if unfreeze == "text":
            text.requires_grad_(True)
        tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=torch.device("cpu"),
                        **_MODE_KW[unfreeze])
        for r in ([10, 11, 0, 1] if unfreeze == "accum" else range(world)):
            b = _batch(r)
            loss, _, _ = tr.losses(b["pixels"], b["latents"], b["noise"], b["t"], b["ids"], b["pidx"])
            loss.backward()
        tr.world = world
        tr.clip_grad_norm()            # (no-op without max_grad_norm; on the averaged gradient with it)
        tr.optimizer_step()
        # Adam's first step moves a coordinate by ~lr*g/|g|: where g is at rounding-noise level the summation order
        # (all-reduce vs in-place accumulation) may flip it.  Everything else must agree to fp32 rounding.
        diff = (p0 - tr.flat.data).abs()
        assert float(diff.max()) <= 2.2e-3
        assert float((diff > 3e-6).float().mean()) < 1e-4      # both legs run 2 intra-op threads (the BLAS partitioning decides which noise-level coordinates flip)
    finally:
        torch.set_num_threads(old_threads)
        ops.set_backend(old_b)
        ops.ACT = old_act


def _order_worker(rank, world, port, out_dir, tuning):
    """1-rank group with E4T_FORCE_COMM=1: record when each gradient region's all-reduce is enqueued."""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", E4T_FORCE_COMM="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(2)
    from e4t.trainer import E4TTrainer
    _, _, n_unet, n_enc, text = _build()
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long),
                    device=torch.device("cpu"), tuning=tuning, max_grad_norm=1.0 if tuning else None)
    assert tr._comm and tr.regions is not None
    log, marks = [], []
    orig = tr._reduce_region

    def spy(key, force=False):
        if key not in tr._done and (tr._armed or force):
            log.append((key, bool(force), len(marks)))
        return orig(key, force)
    tr._reduce_region = spy
    # marker: the backward of the ENCODER pass starts when the gradient of the mid-block output of that pass arrives
    real_forward = n_unet.forward

    def fwd(*a, **k):
        out = real_forward(*a, **k)
        if k.get("return_encoder_outputs") and torch.is_grad_enabled():
            out["down_block_samples"][-1].register_hook(lambda g: marks.append("encoder_pass_backward_started"))
        return out
    n_unet.forward = fwd
    b = _batch(0)
    tr.train_step(b["pixels"], b["ids"], b["pidx"], noise=b["noise"], timesteps=b["t"], latents=b["latents"])
    torch.save(dict(log=log, regions=tr.regions, numel=tr.flat.numel), os.path.join(out_dir, "order.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("tuning", [False, True], ids=["pretrain", "tuning"])
def test_gradient_regions_are_reduced_as_soon_as_they_are_final(tmp_path, tuning):
    """SURVEY §8e bucket schedule: U (up blocks) is enqueued when the full-pass backward leaves the up blocks, H (E4T encoder)
    before the encoder-pass UNet backward starts, D (mid/down, shared by both passes) last — none of them by the
    post-backward sweep, in pre-training and in tuning (where every UNet parameter is in U / D)."""
    mp.spawn(_order_worker, args=(1, _free_port(), str(tmp_path), tuning), nprocs=1, join=True)
    r = torch.load(tmp_path / "order.pt")
    keys = [k for k, _, _ in r["log"]]
    assert keys == ["U", "H", "D"], r["log"]          # W (the head's stacked weights) is exchanged as factors by the head's backward: no all-reduce
    assert not any(forced for _, forced, _ in r["log"]), r["log"]
    by = {k: m for k, _, m in r["log"]}
    assert by["U"] == 0 and by["H"] == 0 and by["D"] == 1, r["log"]       # U and H before the encoder-pass backward, D after
    (w0, w1), (h0, h1), (d0, d1), (u0, u1) = r["regions"]["W"], r["regions"]["H"], r["regions"]["D"], r["regions"]["U"]
    assert w0 == 0 and w1 == h0 and h1 == d0 and d1 == u0 and u1 == r["numel"] and u1 > u0 > d0 > h0 > 0


def _diverge_worker(rank, world, port, out_dir):
    """replicas that DO diverge (rank 1's parameters nudged behind the trainer's back after the start-up broadcast) must abort"""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), E4T_REPLICA_CHECK_EVERY="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from e4t.trainer import E4TTrainer
    _, _, n_unet, n_enc, text = _build()
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=torch.device("cpu"))
    if rank == 1:
        tr.flat.data[5] += 1e-3
    b = _batch(rank)
    try:
        tr.train_step(b["pixels"], b["ids"], b["pidx"], noise=b["noise"], timesteps=b["t"], latents=b["latents"])
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"msg{rank}.txt"), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_diverged_replicas_abort(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_diverge_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert "replicas diverged" in open(tmp_path / f"msg{r}.txt").read()
