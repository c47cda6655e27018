"""Weight-offset heads — drop-in for the reference's ``e4t/weightoffsets.py`` (class :5-23).

Same constructor, same parameter names (``v, linear1, linear2, linear_column, linear_row`` — the
checkpoint contract of ``weight_offsets.pt``), same ``forward()`` result (the (column_dim, row_dim)
offset matrix), but evaluated in closed form by the grouped HIP kernels of csrc/wo.hip instead of
two dense GEMMs per call.  On the hot path the offsets are never materialised on their own:
``WOBank`` evaluates ``W_eff = W o (1 + offsets)`` for every attention projection of a UNet section
in one launch, once per optimisation step, and both UNet passes consume the cached bf16 ``W_eff``.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import _C, ops
from .functional import WOBankFn, WOSlot

f32 = torch.float32
_PNAMES = ("v", "w1", "b1", "w2", "b2", "wc", "bc", "wr", "br")


def _wo_params(wo: "WeightOffsets") -> Dict[str, torch.Tensor]:
    return dict(v=wo.v, w1=wo.linear1.weight, b1=wo.linear1.bias, w2=wo.linear2.weight, b2=wo.linear2.bias,
                wc=wo.linear_column.weight, bc=wo.linear_column.bias, wr=wo.linear_row.weight, br=wo.linear_row.bias)


class _OffsetsFn(torch.autograd.Function):
    """Standalone WeightOffsets.forward(): offsets (col,row) fp32 through the same grouped kernels."""

    @staticmethod
    def forward(ctx, wo, *params):
        dev = params[0].device
        p = dict(zip(_PNAMES, (t.detach() for t in params)))
        out = torch.empty((wo.column_dim, wo.row_dim), dtype=f32, device=dev)
        ones = torch.ones((wo.column_dim, wo.row_dim), dtype=f32, device=dev)
        e = ops.WOEntry(row=wo.row_dim, col=wo.column_dim, W=ones, params=p, weff=out,
                        mode=_C.WO_STORE_F32 | _C.WO_OFFSETS_ONLY)
        ops.backend().wo_forward(ops.WOTable([e]))
        ctx.entry = e
        return out

    @staticmethod
    def backward(ctx, g):
        e = ctx.entry
        e.dweff = g.contiguous().float()
        e.grads = {"g_" + k: torch.empty_like(v) for k, v in e.params.items()}
        ops.backend().wo_backward(ops.WOTable([e]), False)
        return (None, *[e.grads["g_" + k] for k in _PNAMES])


class WeightOffsets(nn.Module):
    def __init__(self, row_dim: int, column_dim: int):
        super().__init__()
        self.row_dim, self.column_dim = row_dim, column_dim
        self.v = nn.Parameter(torch.ones(1))
        self.linear1 = nn.Linear(1, row_dim)
        self.linear2 = nn.Linear(1, column_dim)
        self.linear_column = nn.Linear(row_dim, row_dim)
        self.linear_row = nn.Linear(column_dim, column_dim)

    def forward(self) -> torch.Tensor:
        p = _wo_params(self)
        return _OffsetsFn.apply(self, *[p[k] for k in _PNAMES])


class WOBank:
    """All weight-offset heads of one UNet section (e.g. the up blocks), evaluated and differentiated
    with grouped launches.  ``add()`` registers one projection group sharing an input (fused q|k|v of a
    self-attention, q of a cross-attention, fused k|v of a cross-attention) and returns its WOSlot."""

    def __init__(self, name: str):
        self.name = name
        self._groups: List[dict] = []
        self.table: Optional[ops.WOTable] = None
        self.slots: List[WOSlot] = []
        self._token = None
        self._token_used = False
        self._key = None
        self._root = None
        self.on_backward_done = None

    # -- construction -----------------------------------------------------------------------
    def add(self, linears: List[nn.Linear], wos: List[WeightOffsets]) -> int:
        self._groups.append(dict(linears=linears, wos=wos))
        return len(self._groups) - 1

    def _build(self, device):
        entries, self.slots = [], []
        for g in self._groups:
            row = g["linears"][0].in_features
            cols = [l.out_features for l in g["linears"]]
            tot = sum(cols)
            weff = torch.empty((tot, row), dtype=ops.ACT, device=device)
            weffT = torch.empty((row, tot), dtype=ops.ACT, device=device)
            dweff = torch.zeros((tot, row), dtype=f32, device=device)
            self.slots.append(WOSlot(weff, weffT, dweff))
            off = 0
            for lin, wo, c in zip(g["linears"], g["wos"], cols):
                p = {k: v.data for k, v in _wo_params(wo).items()}
                grads = {"g_" + k: None for k in p}
                entries.append(ops.WOEntry(row=row, col=c, W=lin.weight.data, params=p, weff=weff[off:off + c],
                                           weffT=weffT[:, off:off + c], dweff=dweff[off:off + c], grads=grads))
                entries[-1]._plist = _wo_params(wo)      # the live nn.Parameters (for .grad hand-off)
                entries[-1]._lin = lin
                off += c
        self.table = ops.WOTable(entries)
        self._device = device

    # -- per-forward ------------------------------------------------------------------------
    def _state_key(self):
        k = ops.weights_epoch()
        for e in self.table.entries:
            k += e._plist["v"]._version + e._plist["wc"]._version + e._lin.weight._version
        return k

    def begin(self, device):
        """Called once at the start of every UNet forward.  Returns the autograd token the bank's
        projections attach to (None under no_grad)."""
        if self.table is None or self._device != device or any(
                e.W.data_ptr() != e._lin.weight.data_ptr() or e.params["v"].data_ptr() != e._plist["v"].data_ptr()
                for e in (self.table.entries[0], self.table.entries[-1])):
            self._build(device)
            self._key = None
        key = self._state_key()
        stale = key != self._key
        if not torch.is_grad_enabled():
            if stale:
                self._run_forward()
                self._key = key
            self._token = None
            return None
        if stale or self._token is None or self._token_used:
            if self._root is None or self._root.device != device:
                self._root = torch.zeros(1, dtype=f32, device=device, requires_grad=True)
            self._skip_forward = not stale
            self._token = WOBankFn.apply(self._root, self)
            self._token_used = False
            self._key = key
        return self._token

    def _run_forward(self):
        if getattr(self, "_skip_forward", False):
            self._skip_forward = False
            return
        ops.backend().wo_forward(self.table)

    def _run_backward(self):
        self._token_used = True
        ents = self.table.entries
        # hand the kernels persistent gradient storage: adopt an existing .grad (e.g. a flat-buffer view
        # installed by the trainer), else allocate once and keep.
        # The kernels always accumulate (+=): a parameter whose .grad is None gets a zeroed buffer.
        accumulate = True
        for e in ents:
            for k, p in e._plist.items():
                g = e.grads["g_" + k]
                if p.grad is None:
                    if g is None:
                        e.grads["g_" + k] = torch.zeros_like(p.data)
                    else:
                        g.zero_()
                elif g is None or g.data_ptr() != p.grad.data_ptr():
                    e.grads["g_" + k] = p.grad.data
            e.g_W = None
            if e._lin.weight.requires_grad:
                if e._lin.weight.grad is None:
                    buf = getattr(e, "_gW_buf", None)
                    if buf is None:
                        buf = e._gW_buf = torch.zeros_like(e._lin.weight.data)
                    else:
                        buf.zero_()
                    e.g_W = buf
                else:
                    e.g_W = e._lin.weight.grad.data
        # a slot nobody consumed this step contributes zero
        for s in self.slots:
            if not s.dweff_valid:
                s.dweff.zero_()
        ops.backend().wo_backward(self.table, accumulate)
        for e in ents:
            for k, p in e._plist.items():
                if p.grad is None:
                    p.grad = e.grads["g_" + k]
            if e.g_W is not None and e._lin.weight.grad is None:
                e._lin.weight.grad = e.g_W
        for s in self.slots:
            s.dweff_valid = False
        if self.on_backward_done is not None:
            self.on_backward_done(self)          # e.g. the trainer launches this bank's gradient all-reduce now
