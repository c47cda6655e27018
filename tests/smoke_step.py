"""The check behind __graft_entry__.smoke() (kept under tests/: it needs the oracle, which product code must not import).
One small invocation of the whole hot path on cuda:0 through the HIP kernels, checked against
the CPU fp32 oracle (the oracle is only the checker here; see oracle/e4t_oracle.py header)."""
from __future__ import annotations

import torch

TINY_VIT = dict(image_size=28, patch_size=14, width=128, layers=2, heads=2, mlp_ratio=4.0)
BOC = (64, 128, 128, 128)
TEXT_CFG = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=128, max_len=9, act="quick_gelu")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def run(dev, verbose=True):
    """tiny SD-1 topology, B=2: oracle (CPU fp32) vs native (HIP, bf16) vs stock autocast(bf16) of the oracle — every
    quantity within 2x the stock-autocast error (tests/parity_step.py), plus the absolute round-1 bounds."""
    import parity_step
    from e4t import ops
    assert isinstance(ops.backend(), ops.HipBackend), "smoke must run on the HIP backend"
    rep = parity_step.run("tiny_sd1", dev, verbose=verbose)
    assert rep["n_bad"] == 0, rep
    errs = dict(enc_maps=rep["enc_maps"]["worst"]["native"], losses=rep["losses"]["worst"]["native"], worst_grad=rep["grads"]["worst"]["native"],
                worst_grad_autocast=rep["grads"]["worst"]["autocast"], tightest=rep["grads"]["tightest"]["used"])
    assert errs["enc_maps"] < 3e-2 and errs["losses"] < 2e-2, errs
    print("smoke ok")
    return errs
