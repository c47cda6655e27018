"""BASELINE configs[4] (SD-2.1 @ 768 px, v-prediction) training steps at a small per-GPU batch, eager or replayed from the step graph —
the subject of tools/idle_report.py when the question is "host-bound or device-bound at B = 1".
usage: c5_step.py [B] [graph|eager] [steps]"""
import sys, time, torch
sys.path[:0] = ["e4t-diffusion_amd", "."]
import bench
from e4t.trainer import E4TTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
graph = (sys.argv[2] if len(sys.argv) > 2 else "eager") == "graph"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, "sd21", seed=0)
empty_ids = torch.tensor([[49406] + [49407] * 76], device=dev)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, prediction_type="v_prediction", class_token_id=1125, empty_prompt_ids=empty_ids, device=dev)
if graph:
    assert tr.enable_step_graph()
gen = torch.Generator(device=dev).manual_seed(1)
px = torch.rand((B, 3, 768, 768), generator=gen, device=dev) * 2 - 1
ids = torch.randint(0, 49000, (B, 77), generator=gen, device=dev); pidx = torch.randint(1, 20, (B,), generator=gen, device=dev)
for _ in range(4):
    tr.train_step(px, ids, pidx)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps):
    tr.train_step(px, ids, pidx)
host = time.perf_counter() - t
torch.cuda.synchronize()
print("C5 B=%d %s: %.2f ms/step (host enqueue %.2f ms/step)" % (B, "graph" if graph else "eager", (time.perf_counter() - t) / steps * 1e3, host / steps * 1e3))
