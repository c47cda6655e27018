#!/bin/bash
# round 4, GPU call G: attention checks after the spill fixes, dh-64 dK/dV occupancy A/B on C5, the whole -m gpu suite with durations, smoke
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tests/gpu_report.py attention norms > $O/r04g_checks.txt 2>&1; stamp "checks rc=$?"; grep -c "\[ok\]" $O/r04g_checks.txt; grep "FAIL\|TOTAL" $O/r04g_checks.txt | head
for occ in 2 3; do
E4T_ATTN_DKV_OCC=$occ timeout 600 python - > $O/r04g_c5_occ$occ.txt 2>&1 <<'PY'
import sys, time, torch
sys.path[:0] = ["e4t-diffusion_amd", "."]
import bench
from e4t.trainer import E4TTrainer
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, "sd21", seed=0)
empty_ids = torch.tensor([[49406] + [49407] * 76], device=dev)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, prediction_type="v_prediction", class_token_id=1125, empty_prompt_ids=empty_ids, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
for B in (1, 4):
    px = torch.rand((B, 3, 768, 768), generator=gen, device=dev) * 2 - 1
    ids = torch.randint(0, 49000, (B, 77), generator=gen, device=dev); pidx = torch.randint(1, 20, (B,), generator=gen, device=dev)
    for _ in range(3): tr.train_step(px, ids, pidx)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(6): tr.train_step(px, ids, pidx)
    torch.cuda.synchronize(); print("C5 B=%d %.2f ms/step" % (B, (time.perf_counter() - t) / 6 * 1e3))
PY
stamp "C5 occ=$occ"; grep "C5 B" $O/r04g_c5_occ$occ.txt
done
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $O/r04g_gpu_tests.txt 2>&1; stamp "pytest -m gpu rc=$?"; tail -50 $O/r04g_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04g_smoke.txt 2>&1; stamp "smoke rc=$?"; tail -2 $O/r04g_smoke.txt
stamp done
