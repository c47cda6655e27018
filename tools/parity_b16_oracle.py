"""Direct oracle comparison of the step bench.py times (BASELINE configs[1]: full SD-1.4 UNet + ViT-H-14 + CLIP-L text + VAE encoder, 512 px,
B = 16) — TEST INFRASTRUCTURE, run offline once per round on the GPU box (the two CPU fp32 oracle steps at B = 16 take minutes, which is why
this is not a `-m gpu` test; the suite's own B = 16 evidence is test_full_sd14_batch16_step_matches_sixteen_single_sample_steps).

    E4T_COMMIT=<sha> python tools/parity_b16_oracle.py [--batch 16] [--out profiles/r06_parity/parity_full_sd14_b16_oracle.json]

Protocol = tests/parity_step.py: native B-step on the HIP kernels vs the CPU fp32 oracle B-step, every compared quantity
bounded by 2 x (stock torch.autocast(bf16) of the oracle vs the oracle) + 3e-3, LeakyReLU kink elements inside the measured band aligned per
leg (parity_step.evaluate, the function the -m gpu suite calls at B = 1).
"""
import argparse
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "e4t-diffusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--case", default="full_sd14")
    ap.add_argument("--one-cpu-leg", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_full_sd14_b16_oracle.json"))
    args = ap.parse_args()
    import parity_step as ps
    torch.set_num_threads(args.threads or min(os.cpu_count() or 8, 128))
    dev = torch.device("cuda:0")
    case = dataclasses.replace(ps.cases()[args.case], B=args.batch)
    sec = {}
    t = time.perf_counter()
    o = ps.build_oracle(case)
    n = ps.build_native(case, o, dev)
    d = ps.make_data(case)
    sec["build"] = time.perf_counter() - t
    t = time.perf_counter()
    nat = ps.native_leg(case, n, d, dev)
    del n
    torch.cuda.empty_cache()
    sec["native"] = time.perf_counter() - t
    # exactly the suite's protocol (parity_step.evaluate: stock-autocast calibration leg on the GPU, one CPU fp32 oracle run aligned to
    # its LeakyReLU branches, one aligned to the native leg's, kink band = KINK_SIGMA x the measured input error), at batch B
    case = dataclasses.replace(case, cpu_calib=False)
    rep, _ = ps.evaluate(case, o, d, nat, dev, verbose=True, strict=False, timings=sec, need_ref=False)
    rep.update(batch=args.batch, seconds=sec, cpu_legs=2, threads=torch.get_num_threads(),
               commit=os.environ.get("E4T_COMMIT", "unknown (set E4T_COMMIT: the GPU box has no .git)"), kink_sigma=ps.KINK_SIGMA)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k not in ("bad",)}, indent=1)[:3000])
    print("n_bad", rep["n_bad"], rep["bad"][:8])
    return 0 if rep["n_bad"] == 0 else 3


if __name__ == "__main__":
    sys.exit(main())
