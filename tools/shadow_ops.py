"""GPU diagnostic: run a parity case's native leg with every op call shadowed by the fp32 restatement (tests/emu_backend.py) on
the SAME inputs, and report the per-call relative error of every output — finds the kernel that loses precision on real data."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
import parity_step as ps
from e4t import ops
from emu_backend import EmuBackend

SKIP = {"workspace", "wo_forward", "wo_backward", "weight_prepare", "conv_weight_prepare", "probe_mfma", "_pack_table", "_timed", "_tile", "_colstats_buf"}


def sclone(t):
    if not torch.is_tensor(t):
        return t
    if t.is_contiguous() or 0 in t.stride():
        return t.clone()
    need = 1 + sum((s - 1) * st for s, st in zip(t.shape, t.stride()))
    buf = torch.empty(need, dtype=t.dtype, device=t.device)
    v = torch.as_strided(buf, t.shape, t.stride())
    v.copy_(t)
    return v


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


class Shadow:
    name = "hip"

    def __init__(self, hip, emu):
        self.hip, self.emu, self.rec = hip, emu, []

    def __getattr__(self, name):
        h = getattr(self.hip, name)
        if name in SKIP or not callable(h) or name.startswith("_"):
            return h
        e = getattr(self.emu, name, None)
        if e is None:
            return h

        def call(*a, **k):
            a2 = [sclone(x) for x in a]
            k2 = {kk: sclone(v) for kk, v in k.items()}
            rh = h(*a, **k)
            try:
                re_ = e(*a2, **k2)
            except Exception as ex:          # noqa
                self.rec.append((name, "EMU-FAIL " + str(ex)[:80], 0.0))
                return rh
            outs_h = list(rh) if isinstance(rh, (tuple, list)) else [rh]
            outs_e = list(re_) if isinstance(re_, (tuple, list)) else [re_]
            shp = " ".join(str(tuple(x.shape)) for x in a if torch.is_tensor(x))[:90]
            extra = " ".join(f"{kk}={v}" for kk, v in k.items() if isinstance(v, (int, float, bool)) and v)
            for i, (x, y) in enumerate(zip(outs_h, outs_e)):
                if torch.is_tensor(x) and torch.is_tensor(y) and x.is_floating_point() and x.shape == y.shape:
                    self.rec.append((name, f"out{i} {shp} {extra}", rel(x, y)))
            for i, (x, y) in enumerate(zip(list(a) + list(k.values()), a2 + list(k2.values()))):
                if torch.is_tensor(x) and torch.is_tensor(y) and x.is_floating_point() and x.shape == y.shape and x.data_ptr() not in [o.data_ptr() for o in outs_h if torch.is_tensor(o)]:
                    r_ = rel(x, y)
                    if r_ > 0:
                        self.rec.append((name, f"arg{i} {shp} {extra}", r_))
            return rh
        return call


dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    case = ps.cases()[name]
    o = ps.build_oracle(case)
    d = ps.make_data(case)
    n = ps.build_native(case, o, dev)
    sh = Shadow(ops.backend(), EmuBackend(round_bf16=True))
    old = ops.set_backend(sh)
    ps.native_leg(case, n, d, dev)
    ops.set_backend(old)
    per = collections.defaultdict(list)
    for op, desc, r in sh.rec:
        per[op].append((r, desc))
    print(f"== {name}: {len(sh.rec)} shadowed outputs")
    for op, lst in sorted(per.items(), key=lambda kv: -max(x[0] for x in kv[1])):
        lst.sort(reverse=True)
        med = sorted(x[0] for x in lst)[len(lst) // 2]
        print(f"  {op:<18s} n={len(lst):<5d} median {med:.2e}  worst: " + " | ".join(f"{r:.2e} {dsc}" for r, dsc in lst[:3]))
