"""Generate golden vectors from the REAL reference (runs only in the build container).

The only reference file that imports without diffusers/kornia/open_clip is e4t/weightoffsets.py
(SURVEY.md §8c).  This script imports it from /root/reference, seeds it, and stores parameters,
forward output and all nine parameter gradients for a random upstream gradient.  The fixtures are
committed; the GPU box never needs /root/reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import torch

sys.path.insert(0, "/root/reference")
from e4t.weightoffsets import WeightOffsets  # noqa: E402  (the reference's own class)

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("sq32", 32, 32, 0), ("r32c16", 32, 16, 1), ("r48c80", 48, 80, 2), ("r160c96", 160, 96, 3)]


def main():
    for name, row, col, seed in CASES:
        torch.manual_seed(seed)
        m = WeightOffsets(row, col).double()
        with torch.no_grad():
            m.v.fill_(0.75 + 0.1 * seed)          # move v off its init so dv terms are exercised
        out = m()
        g = torch.randn(col, row, dtype=torch.float64, generator=torch.Generator().manual_seed(100 + seed))
        out.backward(g)
        blob = {
            "row": row, "col": col,
            "params": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "out": out.detach().clone(),
            "upstream": g,
            "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters()},
        }
        torch.save(blob, os.path.join(HERE, f"weightoffsets_{name}.pt"))
        print(name, tuple(out.shape), float(out.abs().mean()))


if __name__ == "__main__":
    main()
