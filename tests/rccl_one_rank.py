"""Launched by tests/test_rccl_gpu.py under torch.distributed.run with ONE rank on a real MI355X (no 8-GPU node is available to
the build): the data-parallel path of the trainer on real RCCL — hook-triggered async all-reduces of the three gradient
regions on RCCL's stream, the waits before the clip / AdamW — must leave exactly the parameters of the no-communication run
(a 1-rank sum is the identity, so any difference is a stream-ordering bug: a region reduced before its last gradient kernel
finished, or AdamW reading a region before its all-reduce completed)."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, HERE, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(dev, comm, tuning, steps=3, collectives="torch"):
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    os.environ["E4T_FORCE_COMM"] = "1" if comm else "0"
    _, _, n_unet, n_enc, text_t = build(seed=0)
    text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
    text.load_state_dict(text_t.state_dict())
    n_unet.to(dev), n_enc.to(dev), text.to(dev)
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev),
                    device=dev, tuning=tuning, max_grad_norm=1.0 if tuning else None, collectives=collectives)
    assert tr._comm == comm and (tr.regions is not None) == comm and (tr._lib_comm is not None) == (comm and collectives == "library")
    log = []
    if comm:
        orig = tr._reduce_region

        def spy(key, force=False):
            if key not in tr._done and (tr._armed or force):
                log.append((key, bool(force)))
            return orig(key, force)
        tr._reduce_region = spy
    g = torch.Generator().manual_seed(3)
    B = 2
    pidx = torch.tensor([2, 4], device=dev)
    losses = []
    for s in range(steps):
        px, lat = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g) * 0.18215
        noise, t, ids = torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randint(1, 99, (B, 9), generator=g)
        kw = dict(noise=noise.to(dev), timesteps=t.to(dev), latents=lat.to(dev))
        if s == 1:      # a gradient-accumulation micro-batch: no collective, no optimiser step
            out = tr.train_step(px.to(dev), ids.to(dev), pidx, sync=False, loss_scale=0.5, **kw)
        else:
            out = tr.train_step(px.to(dev), ids.to(dev), pidx, loss_scale=0.5 if s == 2 else 1.0, **kw)
        losses.append(torch.stack([o.detach().float() for o in out]).cpu())
    torch.cuda.synchronize()
    if comm:        # the trainer probed which stream RCCL's collectives overlap with and moved the step there if the caller's does not
        probe = tr.stream_probe
        assert len(probe) == 6 and (tr._train_stream is None) == (probe[0] or not any(probe)), (probe, tr._train_stream)
        where = "the caller's stream" if tr._train_stream is None else "a stream of the trainer's"
        print(f"rccl one-rank: collectives run beside [caller's stream, 5 new streams] = {probe}; the step ran on {where}")
    return torch.stack(losses), tr.flat.data.detach().cpu().clone(), log


def run_graph(dev, comm, graph, steps=4, hop=False):
    """plain synchronising steps; graph=True: the whole step INCLUDING the regions' all-reduces replayed from one HIP graph"""
    from test_train_step_host_logic import TEXT_CFG, build
    from e4t.text import CLIPTextModel
    from e4t.trainer import E4TTrainer
    os.environ["E4T_FORCE_COMM"] = "1" if comm else "0"
    _, _, n_unet, n_enc, text_t = build(seed=0)
    text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
    text.load_state_dict(text_t.state_dict())
    n_unet.to(dev), n_enc.to(dev), text.to(dev)
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long, device=dev), device=dev)
    assert tr._comm == comm
    if hop:         # as if the probe had found the caller's stream on RCCL's hardware queue: the step runs on a stream of the trainer's
        tr._train_stream, tr._train_stream_probed = torch.cuda.Stream(device=dev), True
    if graph:
        assert tr.enable_step_graph(True)
    g = torch.Generator().manual_seed(5)
    B = 2
    pidx = torch.tensor([2, 4], device=dev)
    losses = []
    for s in range(steps):
        px, lat = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, torch.randn(B, 4, 16, 16, generator=g) * 0.18215
        noise, t, ids = torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g), torch.randint(1, 99, (B, 9), generator=g)
        out = tr.train_step(px.to(dev), ids.to(dev), pidx, noise=noise.to(dev), timesteps=t.to(dev), latents=lat.to(dev))
        losses.append(torch.stack([o.detach().float() for o in out]).cpu())
    torch.cuda.synchronize()
    return torch.stack(losses), tr.flat.data.detach().cpu().clone(), (len(tr._step_graphs), tr._graph_failed)


def library_comm(dev):
    """the C ABI's own communicator (include/e4t_hip.h e4t_comm_*): the collectives themselves, their ordering against the compute
    stream, and the trainer with collectives="library" against the no-communication run"""
    from e4t.comm import LibraryComm
    from e4t import _C
    c = LibraryComm.from_process_group(None)
    assert (c.rank, c.world) == (0, 1) and c.stream_handle() != 0
    # ordering: the buffer's producer is still running on the compute stream when the all-reduce is issued; the consumer is queued after wait()
    x = torch.zeros(1 << 24, device=dev)
    torch.cuda._sleep(20_000_000)
    x.add_(3.0)
    h = c.all_reduce(x)                                    # 1 rank: sum = identity
    h.wait()
    y = x * 2
    assert float(y.min()) == 6.0 and float(y.max()) == 6.0
    for op in ("avg", "min", "max"):
        c.all_reduce(x, op).wait()
    assert float(x.min()) == 3.0 and float(x.max()) == 3.0
    b = torch.randn(5, 1280, device=dev).bfloat16()
    out = torch.empty(5, 1280, device=dev, dtype=torch.bfloat16)
    c.all_gather_into_tensor(out, b).wait()
    assert torch.equal(out, b)
    for bad, kw in ((torch.zeros(4, device=dev, dtype=torch.int32), {}), (torch.zeros(4), {}), (x[::2], {})):
        try:
            c.all_reduce(bad, **kw)
            raise AssertionError("accepted")
        except ValueError:
            pass
    lib = _C.load()
    assert lib.e4t_comm_allreduce(c._h, x.data_ptr(), 4, 7, 0, None) == -22 and b"dtype" in lib.e4t_last_error()
    assert lib.e4t_comm_allreduce(None, x.data_ptr(), 4, 0, 0, None) == -22
    c.close()
    try:
        c.all_reduce(x)
        raise AssertionError("closed communicator accepted a collective")
    except _C.E4TError:
        pass
    print("rccl one-rank, library communicator: all-reduce / all-gather ordered against the compute stream, argument errors refused")
    for tuning in (False, True):
        l0, p0, _ = run(dev, comm=False, tuning=tuning)
        l1, p1, log = run(dev, comm=True, tuning=tuning, collectives="library")
        assert torch.equal(l0, l1), (tuning, l0, l1)
        assert torch.equal(p0, p1), (tuning, float((p0 - p1).abs().max()))
        assert log == [("U", False), ("H", False), ("D", False), ("U", False), ("W", False), ("H", False), ("D", False)], log
        print(f"rccl one-rank {'tuning' if tuning else 'pretrain'}, collectives='library': {p0.numel()} parameters bit-identical with / without the collective path")


def main():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    for tuning in (False, True):
        l0, p0, _ = run(dev, comm=False, tuning=tuning)
        l1, p1, log = run(dev, comm=True, tuning=tuning)
        assert torch.equal(l0, l1), (tuning, l0, l1)
        assert torch.equal(p0, p1), (tuning, float((p0 - p1).abs().max()))
        # two synchronising steps, each: U, H announced by the backward hooks, D by the mid/down bank — none by the sweep.  W (the head's
        # stacked weight gradient) crosses as gathered factors in the first (RCCL all-gather, no all-reduce) and rides the all-reduce in
        # the second, whose stack already holds the accumulated micro-batch
        assert log == [("U", False), ("H", False), ("D", False), ("U", False), ("W", False), ("H", False), ("D", False)], log
        print(f"rccl one-rank {'tuning' if tuning else 'pretrain'}: {p0.numel()} parameters bit-identical with / without the collective path; regions {log[:3]}")
    # the step graph under a communicator (BASELINE configs[4] is one image per GPU on eight GPUs): collectives captured with the step
    l0, p0, _ = run_graph(dev, comm=False, graph=False)
    l1, p1, (n_graphs, failed) = run_graph(dev, comm=True, graph=True)
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
    assert (n_graphs == 1) != failed, (n_graphs, failed)
    print(f"rccl one-rank step graph with the collectives inside: {'replayed, ' if n_graphs else 'capture refused -> eager fallback, '}bit-identical to the eager no-comm run")
    l1, p1, _ = run_graph(dev, comm=True, graph=False, hop=True)
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
    print("rccl one-rank, the step on a stream of the trainer's: bit-identical to the eager no-comm run")
    library_comm(dev)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")


if __name__ == "__main__":
    main()
