"""GPU: which torch (aten) device kernels remain in one training step, from which source line, and what do they cost on the device?
(round-4 review, weak #8: fills / adds / copies / cats next to the HIP kernels.)  torch.profiler with stacks over ONE steady-state step.

    python tools/glue_profile.py [sd14|sd21] [B] [tuning]   ->  gpurun_out/glue_profile_<model>_b<B>.txt
"""
import collections
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from e4t.trainer import E4TTrainer  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "sd14"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tuning = len(sys.argv) > 3 and sys.argv[3] == "tuning"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, model, 0)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, empty_prompt_ids=torch.tensor([[49406] + [49407] * 76], device=dev), device=dev,
                prediction_type="epsilon" if model == "sd14" else "v_prediction", tuning=tuning, max_grad_norm=1.0 if tuning else None)
res = 512 if model == "sd14" else 768
g = torch.Generator(device=dev)
g.manual_seed(0)
mk = lambda: (torch.rand((B, 3, res, res), generator=g, device=dev) * 2 - 1, torch.randint(0, 49000, (B, 77), generator=g, device=dev),
              torch.randint(1, 20, (B,), generator=g, device=dev))
pool = [mk() for _ in range(3)]
for i in range(3):
    tr.prefetch(pool[(i + 1) % 3][0])
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.prefetch(pool[1][0])
    tr.train_step(*pool[0])
    torch.cuda.synchronize()

rows = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
n_kern, t_kern = 0, 0.0
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    p, nested = ev.cpu_parent, False
    while p is not None:
        if p.name.startswith("aten::"):
            nested = True
            break
        p = p.cpu_parent
    dt = ev.device_time_total
    if nested or dt <= 0:
        continue
    where = "autograd engine / no python frame"
    for fr in (ev.stack or []):
        if "e4t-diffusion_amd/e4t/" in fr or "/bench.py" in fr:
            where = fr.split("e4t-diffusion_amd/")[-1].strip()
            break
    r = rows[(ev.name, where)]
    r[0] += 1
    r[1] += dt
    shp = ev.input_shapes[0] if ev.input_shapes else None
    r[2][str(shp)] += 1
    n_kern += 1
    t_kern += dt
out = os.path.join(R, "gpurun_out", f"glue_profile_{model}_b{B}{'_tuning' if tuning else ''}.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as fh:
    fh.write(f"# {model} B={B} tuning={tuning}: {n_kern} top-level aten ops with device time in one step, {t_kern / 1e3:.3f} ms of device time\n")
    for (name, where), (n, dt, shapes) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        top = "; ".join(f"{k} x{v}" for k, v in shapes.most_common(3))
        fh.write(f"{n:5d}  {dt:9.1f} us  {name:<28s} {where}   [{top}]\n")
print(open(out).read()[:6000])
