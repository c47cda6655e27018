"""Which freshly created HIP streams run BESIDE the current stream?  (ROCm maps streams onto a few hardware queues; two streams on one
queue serialise.)  Probes 10 successive streams, without and with an initialised 1-rank RCCL process group.
usage: python tools/probe/stream_queues.py [pg]"""
import os
import sys
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if len(sys.argv) > 1:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t)
    w = dist.all_reduce(t, async_op=True); w.wait()
    torch.cuda.synchronize()


def beside(main, s):
    torch.cuda.synchronize()
    e_main, e_side = torch.cuda.Event(), torch.cuda.Event()
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        torch.cuda._sleep(int(5e7))
        e_main.record(main)
    with torch.cuda.stream(s):
        x.add_(1)
        e_side.record(s)
    ok = False
    while not e_main.query():
        if e_side.query():
            ok = True
            break
    torch.cuda.synchronize()
    return ok


main = torch.cuda.current_stream(dev)
keep = []
for i in range(10):
    s = torch.cuda.Stream(device=dev)
    keep.append(s)
    print(f"{'pg ' if len(sys.argv) > 1 else 'plain '}stream #{i} ({s.stream_id}): runs beside the current stream: {beside(main, s)}", flush=True)

if len(sys.argv) > 1:
    # ... and RCCL's own stream: does a collective run beside a busy training stream, for the current stream and for other candidates?
    # (a collective waits for the stream it is called from, so it is called from an idle third stream)
    src, dst = torch.ones(1 << 20, device=dev), torch.empty(1 << 20, device=dev)
    for name, train in [("current", main)] + [(f"#{i}", s) for i, s in enumerate(keep[:8])]:
        other = next(s for s in keep if s is not train and beside(train, s))
        torch.cuda.synchronize()
        e_train = torch.cuda.Event()
        with torch.cuda.stream(train):
            torch.cuda._sleep(int(5e7))
            e_train.record(train)
        with torch.cuda.stream(other):
            w = dist.all_gather_into_tensor(dst, src, async_op=True)
        ok = False
        while not e_train.query():
            if w.is_completed():
                ok = True
                break
        torch.cuda.synchronize()
        print(f"pg RCCL's stream: a collective runs beside training stream {name}: {ok}", flush=True)
    dist.destroy_process_group()
