"""Run every kernel parity check without stopping at the first failure; print a table.
    python tests/gpu_report.py [family ...]   (GPU box)"""
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "e4t-diffusion_amd"), HERE, os.path.join(ROOT, "oracle")]

import torch  # noqa: E402
from e4t import ops  # noqa: E402
from emu_backend import EmuBackend  # noqa: E402
import kernel_checks as kc  # noqa: E402


def main():
    want = set(sys.argv[1:])
    dev = torch.device("cuda:0")
    hip, emu = ops.HipBackend(), EmuBackend()
    nbad = 0
    for fam, fn in kc.all_checks(hip, emu, dev, ops):
        if want and fam not in want:
            continue
        print(f"== {fam}", flush=True)
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception:
            traceback.print_exc()
            nbad += 1
            continue
        for n, e, t in res:
            ok = e <= t
            nbad += (not ok)
            print(f"  [{'ok' if ok else 'FAIL'}] {n:<60s} rel_l2={e:.3e} tol={t:.1e}", flush=True)
    print(f"TOTAL FAILURES: {nbad}")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
