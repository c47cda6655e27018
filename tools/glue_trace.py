"""Which torch (aten) kernels does a training step launch OUTSIDE the HIP op layer, and from where?  (CPU, op emulation.)

The product's arithmetic goes through e4t.ops.backend(); everything else a step launches is torch glue — fills, adds, copies, cats
(round-4 review: a few hundred such launches per step, ~2.3 ms of device time plus their boundaries).  This runs one tiny training step
with the emulation backend, wraps every backend method so that aten ops INSIDE it are ignored, and records every device-kernel-producing
aten op outside it with the innermost e4t/ source line (or "autograd engine" when no Python frame of the package is on the stack).

    python tools/glue_trace.py [--tuning] [--steps 2]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "e4t-diffusion_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_OPS = ("view", "reshape", "slice", "select", "expand", "permute", "transpose", "t.default", "as_strided", "unsqueeze", "squeeze", "detach", "alias",
            "_unsafe_view", "unbind", "split", "narrow", "empty", "new_empty", "set_", "stride", "size", "is_", "_local_scalar", "lift_fresh", "_to_copy_noop",
            "sym_", "_reshape_alias", "chunk", "result_type", "prim.", "_has_compatible", "empty_like", "empty_strided", "unfold", "view_as", "diagonal")


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.depth = 0
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if self.depth == 0 and not any(v in name for v in VIEW_OPS):
            where = "autograd engine / torch"
            for fr in reversed(traceback.extract_stack()):
                if "/e4t-diffusion_amd/e4t/" in fr.filename:
                    where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            self.ops[(name, where, len(shape) and int(torch.tensor(shape).prod()))] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tuning", action="store_true")
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    from e4t import ops
    from emu_backend import EmuBackend
    from test_train_step_host_logic import build
    from e4t.trainer import E4TTrainer
    be = EmuBackend(round_bf16=True)
    ops.set_backend(be)
    ops.ACT = torch.bfloat16
    rec = Rec()
    for nm in dir(be):
        f = getattr(be, nm)
        if callable(f) and not nm.startswith("_"):
            def wrap(f):
                def g(*x, **k):
                    rec.depth += 1
                    try:
                        return f(*x, **k)
                    finally:
                        rec.depth -= 1
                return g
            setattr(be, nm, wrap(f))
    _, _, n_unet, n_enc, text = build()
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=torch.device("cpu"),
                    tuning=a.tuning, max_grad_norm=1.0 if a.tuning else None)
    B = 2
    g = torch.Generator().manual_seed(7)
    mk = lambda: (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, torch.randint(1, 99, (B, 9), generator=g), torch.tensor([2, 4]),
                  torch.randn(B, 4, 16, 16, generator=g), torch.tensor([5, 700]), None, torch.randn(B, 4, 16, 16, generator=g) * 0.18215)
    tr.train_step(*mk())                       # warm-up: one-time casts
    with rec:
        for _ in range(a.steps):
            tr.train_step(*mk())
    tot = sum(rec.ops.values())
    print(f"# {tot / a.steps:.0f} glue aten ops per step (tiny SD-1 topology: 2 levels fewer than SD-1.4; counts scale with the block count)")
    by_site = collections.Counter()
    for (name, where, n), c in rec.ops.items():
        by_site[(where, name)] += c
    for (where, name), c in sorted(by_site.items(), key=lambda kv: -kv[1]):
        sizes = collections.Counter()
        for (n2, w2, numel), c2 in rec.ops.items():
            if (w2, n2) == (where, name):
                sizes[numel] += c2
        top = ", ".join(f"{k}el x{v / a.steps:.0f}" for k, v in sizes.most_common(4))
        print(f"{c / a.steps:7.1f}  {name:<34s} {where}   [{top}]")


if __name__ == "__main__":
    main()
