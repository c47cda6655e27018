"""GPU diagnostic: for each parity case print the quantities with the largest native / autocast error ratio, next to the error of
the fp32-math-bf16-storage emulation of the same graph (tests/emu_backend.py) — separates kernel precision from graph design."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
import parity_step as ps
from e4t import ops
from emu_backend import EmuBackend
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    case = ps.cases()[name]
    o = ps.build_oracle(case)
    d = ps.make_data(case)
    n = ps.build_native(case, o, dev)
    nat = ps.native_leg(case, n, d, dev)
    old = ops.set_backend(EmuBackend(round_bf16=True))
    n2 = ps.build_native(case, o, dev)
    emu = ps.native_leg(case, n2, d, dev)
    ops.set_backend(old)
    ref = ps.oracle_leg(case, o, d)
    cal = ps.oracle_leg(case, o, d, dev=dev, autocast=True)
    cal_cpu = ps.oracle_leg(case, o, d, dev=torch.device("cpu"), autocast=True) if os.environ.get("CAL_CPU") else None
    rows = []
    for k in ref:
        if k in nat and ref[k].numel() >= 256:
            rows.append((ps.rel(nat[k], ref[k]), ps.rel(cal[k], ref[k]), ps.rel(emu[k], ref[k]), ps.rel(nat[k], emu[k]),
                         ps.rel(cal_cpu[k], ref[k]) if cal_cpu else 0.0, k))
    rows.sort(key=lambda r: -r[0] / (2 * r[1] + ps.FLOOR))
    over = sum(1 for r in rows if r[0] > 2 * r[1] + ps.FLOOR)
    print(f"== {name}: {len(rows)} quantities, {over} over the bound")
    print("   native   autocast  emu      nat-vs-emu autocastCPU  name")
    for r in rows[:25]:
        print(f"   {r[0]:.2e} {r[1]:.2e} {r[2]:.2e} {r[3]:.2e} {r[4]:.2e}  {r[5]}")
