"""Op-level Python surface of the HIP library: torch tensors in, torch tensors out.

Every function here launches hand-written HIP kernels through the C ABI (``_C.py``) on torch's
current stream; torch is used only to own device memory.  The functions dispatch through a
module-level backend object so that *tests* can substitute an instrumented double (tests/
emu_backend.py) to validate the host-side graph logic on a machine without a GPU.  The product
never installs another backend: with no GPU / no built library every call raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
import os
import threading
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _C

bf16 = torch.bfloat16
f32 = torch.float32

# Storage dtype of activations and compute copies of weights.  bf16 in the product (the HIP kernels
# accept nothing else); tests that drive the host logic through the fp32 emulation set it to fp32.
ACT = bf16

_WEIGHTS_EPOCH = 0

# Held while a HIP graph is being captured (CLIP text encoder, sampling loop).  Stream capture runs in "global" error mode: a
# device allocation, a pinned-memory allocation, an event synchronisation or a host-to-device copy issued by ANOTHER host thread
# while the capture is open invalidates it (hipErrorStreamCaptureUnsupported / ...Invalidated).  The only other thread of a rank that
# talks to the GPU is the data loader's prefetch worker (data.py DeviceLoader._produce): it takes this lock around its GPU section,
# so a capture simply pauses it for the few hundred milliseconds it takes.
capture_lock = threading.RLock()


@contextlib.contextmanager
def capture_guard():
    """Everything a HIP graph capture needs from the rest of the process: the loader's prefetch worker paused (capture_lock) and the
    garbage collector off — a cycle holding an OLD graph (a re-captured shape, a previous model) that is collected while a capture
    is open runs ~CUDAGraph -> hipGraphDestroy -> "operation not permitted when stream is capturing" inside a destructor, i.e. a
    process abort (seen when a second trainer was built in one process).  Collects first."""
    was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with capture_lock:
            yield
    finally:
        if was_on:
            gc.enable()


def weights_epoch() -> int:
    """Bumped whenever master weights were updated behind autograd's back (our fused AdamW writes
    through raw pointers, which does not touch tensor._version)."""
    return _WEIGHTS_EPOCH


def bump_weights_epoch():
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_DEV_INDEX = None


def _stream():
    """Raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() costs ~9 us of host time per
    call — 10 ms per training step at ~3500 launches; the raw query is ~0.3 us and still follows `with torch.cuda.stream(...)`.)"""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()      # one process drives one GPU (one rank per device)
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def _rowmajor(t: torch.Tensor, name: str):
    if t.dim() < 2 or t.stride(-1) != 1:
        raise ValueError(f"{name}: expected a row-major matrix with unit inner stride, got strides {t.stride()}")



@dataclass
class WOEntry:
    """One WeightOffsets instance + the projection weight it modulates (row = in, col = out)."""
    row: int
    col: int
    W: torch.Tensor                      # fp32 [col, row]
    params: Optional[dict] = None        # v, w1, b1, w2, b2, wc, bc, wr, br (fp32) or None for a plain weight
    weff: Optional[torch.Tensor] = None  # bf16 [col, >=row] view (row stride = ld)
    weffT: Optional[torch.Tensor] = None # bf16 [row, >=col] view
    dweff: Optional[torch.Tensor] = None # fp32 [col, >=row] view (backward input)
    grads: Optional[dict] = None         # g_v ... g_br fp32 tensors (backward outputs)
    g_W: Optional[torch.Tensor] = None
    vecs: Optional[torch.Tensor] = None
    partial: Optional[torch.Tensor] = None
    mode: int = 0                        # _C.WO_STORE_F32 | _C.WO_OFFSETS_ONLY


@dataclass
class WOTable:
    entries: List[WOEntry]
    dev: Optional[torch.Tensor] = None   # packed e4t_wo_desc array in device memory (HIP backend)
    _sig: tuple = field(default_factory=tuple)

    @property
    def max_row(self):
        return max(e.row for e in self.entries)

    @property
    def max_col(self):
        return max(e.col for e in self.entries)


class HipBackend:
    """The one and only product backend: libe4t_hip.so on the current CUDA(HIP) device."""

    name = "hip"

    def __init__(self):
        self.lib = _C.load()
        self._ws = {}
        self.prof = None   # bench.py: list of (kernel key, algorithmic flops, algorithmic bytes, start event, end event) when enabled

    def _timed(self, key, flops, fn, nbytes=0.0):
        """Run one launch; when profiling is on, bracket it with events on the launch stream."""
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.prof.append((key, flops, float(nbytes() if callable(nbytes) else nbytes), e0, e1))
        return r

    def _plan(self, fn, d, what):
        """tile / split-K / workspace the launcher will use for descriptor `d` (e4t_gemm_plan & co.: the heuristic lives in the
        library only; nothing here mirrors it)"""
        pl = _C.GemmPlan()
        _C.check(fn(C.byref(d), C.byref(pl)), what)
        return pl

    # ------------------------------------------------------------------ workspaces
    def workspace(self, nbytes: int, device) -> torch.Tensor:
        # one scratch buffer per (device, stream): kernels of concurrent streams must not share split-K / stats partials
        key = (device.type, device.index, _stream() if device.type == "cuda" else 0)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    @staticmethod
    def _colstats_buf(out, M, N, wanted):
        """fp32 [M/32, N, 2] buffer for the epilogue's per-32-row column statistics (None when not applicable)."""
        if not wanted or out.dtype != bf16 or M % 32 or not out.is_contiguous():
            return None
        return torch.empty((M // 32, N, 2), dtype=f32, device=out.device)

    # ------------------------------------------------------------------ GEMM
    def gemm(self, a, b, *, a2=None, bias=None, residual=None, rowbias=None, rows_per_batch=0, out=None,
             out_dtype=bf16, gelu=False, accum=False, alpha=1.0, reduce_batch=False, tile=0, splitk=0, colstats=False, panels=None):
        """panels = (rows, stride, offset, count): only the count x rows logical rows that lie `stride` physical rows apart (the first
        at `offset`) in a, out and residual are computed / written — e4t_gemm_desc.panel_*; `out` is required and the other rows of
        it are left untouched."""
        batched = a.dim() == 3
        _rowmajor(a, "gemm A"); _rowmajor(b, "gemm B")
        if batched:
            nb, M, K = a.shape
            N = b.shape[1]
            sA, sB = a.stride(0), b.stride(0)
        else:
            nb, (M, K), N, sA, sB = 1, a.shape, b.shape[0], 0, 0
        if panels is not None:
            pr, ps, po, pc = panels
            assert not batched and out is not None and a.shape[0] >= (pc - 1) * ps + po + pr and out.shape[0] == a.shape[0], "gemm: bad row panels"
            assert residual is None or residual.shape[0] == a.shape[0]
            M = pr * pc
        K1 = K
        if a2 is not None:
            _rowmajor(a2, "gemm A2")
            K = K1 + a2.shape[-1]
        assert b.shape[-1] == K, f"gemm: B has K={b.shape[-1]}, A has K={K}"
        if out is None:
            shape = (M, N) if (not batched or reduce_batch) else (nb, M, N)
            out = torch.empty(shape, dtype=out_dtype, device=a.device)
        _rowmajor(out, "gemm C")
        d = _C.GemmDesc()
        d.A, d.A2, d.B, d.C = _ptr(a), _ptr(a2), _ptr(b), _ptr(out)
        d.bias, d.residual, d.rowbias = _ptr(bias), _ptr(residual), _ptr(rowbias)
        d.M, d.N, d.K, d.K1 = M, N, K, K1
        d.lda, d.lda2, d.ldb, d.ldc = a.stride(-2), (a2.stride(-2) if a2 is not None else 0), b.stride(-2), out.stride(-2)
        d.ldr = residual.stride(-2) if residual is not None else 0
        d.rows_per_batch = rows_per_batch
        d.ldrb = rowbias.stride(0) if rowbias is not None else 0
        flags = 0
        if out.dtype == f32:
            flags |= _C.OUT_F32
        if residual is not None and residual.dtype == f32:
            flags |= _C.RES_F32
        if gelu:
            flags |= _C.ACT_GELU
        if accum:
            flags |= _C.ACCUM
        if reduce_batch:
            flags |= _C.REDUCE_BATCH
        d.flags, d.tile, d.splitk, d.batch, d.alpha = flags, tile, splitk, nb, alpha
        d.strideA, d.strideB = sA, sB
        d.strideC = out.stride(0) if (batched and not reduce_batch) else 0
        d.strideBias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
        if panels is not None:
            d.panel_rows, d.panel_stride, d.panel_off = panels[0], panels[1], panels[2]
        cs = self._colstats_buf(out, M, N, colstats and not batched)
        d.colstats = _ptr(cs)            # before the plan: whether statistics are wanted prices the split-K variants (they leave none)
        pl = self._plan(self.lib.e4t_gemm_plan, d, "e4t_gemm_plan")
        ws = self.workspace(pl.workspace_bytes, a.device) if pl.workspace_bytes else None
        d.workspace, d.workspace_bytes = _ptr(ws), (ws.numel() if ws is not None else 0)
        st = _stream()
        # algorithmic bytes: A, B once; C once (+ once more when read: residual / accumulate)
        nbytes = lambda: (2.0 * nb * M * K + 2.0 * N * K * (nb if sB else 1) + out.element_size() * M * N * (1 if reduce_batch else nb) * (2 if accum else 1)
                          + (residual.element_size() * M * N if residual is not None else 0))
        rc = self._timed(f"gemm{pl.tile}" + (f"s{pl.stages}" if pl.stages else ""), 2.0 * M * N * K * nb,
                         lambda: _C.check(self.lib.e4t_gemm_nt(C.byref(d), st), "e4t_gemm_nt"), nbytes)
        if cs is not None and rc == 1:
            out._e4t_colstats = cs          # consumed by groupnorm_fwd (the GroupNorm of this activation skips its statistics pass)
        return out

    def gemm_tn(self, a, b, *, out=None, out_dtype=f32, accum=False, alpha=1.0, splitk=0):
        """C[M, N] = alpha * a^T . b with a = [K, M], b = [K, N] (bf16, unit inner stride): contraction over the rows —
        dW = dY^T . X without transposes."""
        _rowmajor(a, "gemm_tn A"); _rowmajor(b, "gemm_tn B")
        K, M = a.shape
        N = b.shape[1]
        assert b.shape[0] == K
        if out is None:
            out = torch.empty((M, N), dtype=out_dtype, device=a.device)
        _rowmajor(out, "gemm_tn C")
        d = _C.GemmDesc()
        d.A, d.B, d.C = _ptr(a), _ptr(b), _ptr(out)
        d.M, d.N, d.K, d.K1 = M, N, K, K
        d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
        flags = (_C.OUT_F32 if out.dtype == f32 else 0) | (_C.ACCUM if accum else 0)
        d.flags, d.splitk, d.batch, d.alpha = flags, splitk, 1, alpha
        pl = self._plan(self.lib.e4t_gemm_tn_plan, d, "e4t_gemm_tn_plan")
        ws = self.workspace(pl.workspace_bytes, a.device) if pl.workspace_bytes else None
        d.workspace, d.workspace_bytes = _ptr(ws), (ws.numel() if ws is not None else 0)
        st = _stream()
        self._timed("gemm_tn", 2.0 * M * N * K, lambda: _C.check(self.lib.e4t_gemm_tn(C.byref(d), st), "e4t_gemm_tn"),
                    2.0 * K * (M + N) + out.element_size() * M * N * (2 if accum else 1))
        return out

    # ------------------------------------------------------------------ conv
    def conv3x3(self, x, w, B, Hin, Win, Hout, Wout, mode, *, colstats=False, bias=None, residual=None, rowbias=None, out=None,
                out_dtype=bf16, accum=False, tile=0, splitk=0):
        Cin, Cout = x.shape[-1], w.shape[0]
        assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == 9 * Cin
        M = B * Hout * Wout
        if out is None:
            out = torch.empty((M, Cout), dtype=out_dtype, device=x.device)
        d = _C.ConvDesc()
        d.X, d.W, d.Y, d.bias, d.residual, d.rowbias = _ptr(x), _ptr(w), _ptr(out), _ptr(bias), _ptr(residual), _ptr(rowbias)
        d.B, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = B, Hin, Win, Cin, Hout, Wout, Cout
        flags = 0
        if out.dtype == f32:
            flags |= _C.OUT_F32
        if residual is not None and residual.dtype == f32:
            flags |= _C.RES_F32
        if accum:
            flags |= _C.ACCUM
        d.mode, d.flags, d.tile, d.splitk = mode, flags, tile, splitk
        d.ldrb = rowbias.stride(0) if rowbias is not None else 0
        cs = self._colstats_buf(out, M, Cout, colstats)
        d.colstats = _ptr(cs)            # before the plan (see gemm)
        pl = self._plan(self.lib.e4t_conv3x3_plan, d, "e4t_conv3x3_plan")
        ws = self.workspace(pl.workspace_bytes, x.device) if pl.workspace_bytes else None
        d.workspace, d.workspace_bytes = _ptr(ws), (ws.numel() if ws is not None else 0)
        st = _stream()
        # algorithmic bytes: the input map once (not once per tap), the weights once, the output once (+ residual)
        nbytes = 2.0 * B * Hin * Win * Cin + 2.0 * Cout * 9 * Cin + out.element_size() * M * Cout * (2 if accum else 1) + \
            (residual.element_size() * M * Cout if residual is not None else 0)
        rc = self._timed(f"conv{pl.tile}" + (f"s{pl.stages}" if pl.stages else ""), 2.0 * M * Cout * 9 * Cin,
                         lambda: _C.check(self.lib.e4t_conv3x3(C.byref(d), st), "e4t_conv3x3"), nbytes)
        if cs is not None and rc == 1:
            out._e4t_colstats = cs
        return out

    def conv_weight_prepare(self, w_oihw, Ipad=None, Opad=None, want_fwd=True, want_dgrad=True):
        O, I = w_oihw.shape[:2]
        Ipad = Ipad or (I + 63) // 64 * 64
        Opad = Opad or (O + 63) // 64 * 64
        w = w_oihw.detach().contiguous().float()
        wf = torch.empty((O, 9 * Ipad), dtype=bf16, device=w.device) if want_fwd else None
        wd = torch.empty((I, 9 * Opad), dtype=bf16, device=w.device) if want_dgrad else None
        _C.check(self.lib.e4t_conv_weight_prepare(_ptr(w), _ptr(wf), _ptr(wd), O, I, Ipad, Opad, _stream()), "e4t_conv_weight_prepare")
        return wf, wd

    # ------------------------------------------------------------------ attention
    def attention_fwd(self, q, k, v, B, H, T, S, DH, scale, out=None, need_lse=True, causal=False):
        """q: [B*T, *] view, k/v: [B*S, *] views (row stride arbitrary); returns (o [B*T, H*DH], lse [B,H,T])."""
        for t in (q, k, v):
            _rowmajor(t, "attention operand")
        if out is None:
            out = torch.empty((B * T, H * DH), dtype=bf16, device=q.device)
        lse = torch.empty((B, H, T), dtype=f32, device=q.device) if need_lse else None
        ldq, ldk, ldv, ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        st = _stream()
        self._timed(f"attn_fwd{DH}", 4.0 * B * H * T * S * DH, lambda: _C.check(self.lib.e4t_attention_fwd(
            _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, H, T, S, DH, ldq, ldk, ldv, ldo,
            T * ldq, S * ldk, S * ldv, T * ldo, float(scale), int(causal), st), "e4t_attention_fwd"),
            2.0 * B * H * DH * (2 * T + 2 * S) + 4.0 * B * H * T)
        return out, lse

    def attention_bwd(self, q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, scale, causal=False):
        """dq/dk/dv are pre-allocated views with the SAME strides as q/k/v; do has the strides of o."""
        ldq, ldk, ldv, ldo = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
        assert dq.stride(0) == ldq and dk.stride(0) == ldk and dv.stride(0) == ldv and do.stride(0) == ldo
        # Delta [B][H][T] + (few keys only) the fp32 partials of the query-chunked dK/dV kernel: the library states the size
        nws = self.lib.e4t_attention_bwd_workspace_floats(B, H, T, S, DH)
        ws = torch.empty(nws, dtype=f32, device=q.device)
        st = _stream()
        self._timed(f"attn_bwd{DH}", 10.0 * B * H * T * S * DH, lambda: _C.check(self.lib.e4t_attention_bwd_ws(
            _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(do), _ptr(lse), _ptr(ws), nws, _ptr(dq), _ptr(dk),
            _ptr(dv), B, H, T, S, DH, ldq, ldk, ldv, ldo, T * ldq, S * ldk, S * ldv, T * ldo,
            float(scale), int(causal), st), "e4t_attention_bwd_ws"), 2.0 * B * H * DH * (4 * T + 4 * S) + 8.0 * B * H * T)

    # ------------------------------------------------------------------ norms
    def groupnorm_fwd(self, x1, x2, gamma, beta, B, HW, G, eps, silu):
        C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
        Cn = C1 + C2
        assert x1.is_contiguous() and (x2 is None or x2.is_contiguous())
        stats = torch.empty((B, G, 2), dtype=f32, device=x1.device)
        nb = self.lib.e4t_groupnorm_workspace_bytes(B, HW, Cn, G, 0)
        ws = self.workspace(nb, x1.device)
        y = torch.empty((B * HW, Cn), dtype=bf16, device=x1.device)
        cs1 = getattr(x1, "_e4t_colstats", None)
        cs2 = getattr(x2, "_e4t_colstats", None) if x2 is not None else None
        nblk, nch = HW // 32, self.lib.e4t_groupnorm_num_chunks(B, HW)
        if cs1 is not None and (x2 is None or cs2 is not None) and HW % 32 == 0 and nblk % min(nch, nblk) == 0:
            # the producers left per-32-row column statistics behind: no statistics pass over the activation(s)
            self._timed("gn_fwd_colstats", 0.0, lambda: _C.check(self.lib.e4t_groupnorm_fwd_cs(
                _ptr(x1), C1, _ptr(cs1), _ptr(x2), C2, _ptr(cs2), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats),
                B, HW, G, float(eps), int(silu), _ptr(ws), ws.numel(), _stream()), "e4t_groupnorm_fwd_cs"), 4.0 * B * HW * Cn)
            return y, stats
        self._timed("gn_fwd_2pass", 0.0, lambda: _C.check(self.lib.e4t_groupnorm_fwd(
            _ptr(x1), C1, _ptr(x2), C2, _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), B, HW, G, float(eps),
            int(silu), _ptr(ws), ws.numel(), _stream()), "e4t_groupnorm_fwd"), 6.0 * B * HW * Cn)
        return y, stats

    def groupnorm_fwd_unfused(self, x1, x2, gamma, beta, B, HW, G, eps, silu):
        """stats (+ finalize) and apply as separate ABI calls — kept for the kernel checks of those entry points."""
        C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
        Cn = C1 + C2
        stats = torch.empty((B, G, 2), dtype=f32, device=x1.device)
        ws = self.workspace(self.lib.e4t_groupnorm_workspace_bytes(B, HW, Cn, G, 0), x1.device)
        _C.check(self.lib.e4t_groupnorm_stats(_ptr(x1), C1, _ptr(x2), C2, B, HW, G, float(eps), _ptr(stats), _ptr(ws), ws.numel(), _stream()),
                 "e4t_groupnorm_stats")
        y = torch.empty((B * HW, Cn), dtype=bf16, device=x1.device)
        _C.check(self.lib.e4t_groupnorm_apply(_ptr(x1), C1, _ptr(x2), C2, _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(y), B, HW, G, int(silu), _stream()),
                 "e4t_groupnorm_apply")
        return y, stats

    def groupnorm_bwd(self, x1, x2, dy, stats, gamma, beta, add, B, HW, G, silu, want_param_grads=False, add2=None):
        """add / add2: gradients reaching x1 / x2 through another consumer (bf16, same shapes), fused into the kernel."""
        C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
        Cn = C1 + C2
        assert dy.is_contiguous() and (add is None or (add.is_contiguous() and add.shape == x1.shape and add.dtype == x1.dtype))
        assert add2 is None or (x2 is not None and add2.is_contiguous() and add2.shape == x2.shape and add2.dtype == x2.dtype)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        ch = self.lib.e4t_groupnorm_num_chunks(B, HW)
        cpart = torch.empty((B * ch, Cn, 2), dtype=f32, device=x1.device) if want_param_grads else None
        nb = self.lib.e4t_groupnorm_workspace_bytes(B, HW, Cn, G, 0) + B * G * 8
        ws = self.workspace(nb, x1.device)
        # two passes (statistics of dy.x_hat, then apply): x and dy twice, dx once, the fused residual gradients once
        self._timed("gn_bwd", 0.0, lambda: _C.check(self.lib.e4t_groupnorm_bwd(
            _ptr(x1), C1, _ptr(x2), C2, _ptr(dy), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(add), _ptr(add2), _ptr(dx1),
            _ptr(dx2), _ptr(cpart), B, HW, G, int(silu), _ptr(ws), ws.numel(), _stream()), "e4t_groupnorm_bwd"),
            2.0 * B * HW * (5 * Cn + (C1 if add is not None else 0) + (C2 if add2 is not None else 0)))
        dgamma = dbeta = None
        if want_param_grads:
            s = cpart.sum(dim=0)          # tiny (chunks x C) reduction of kernel partials
            dbeta, dgamma = s[:, 0].contiguous(), s[:, 1].contiguous()
        return dx1, dx2, dgamma, dbeta

    def layernorm_fwd(self, x, gamma, beta, eps, need_stats=True):
        """x: bf16, or fp32 rows of an fp32 residual stream (CLIP-ViT); y is bf16 either way"""
        M, D = x.shape
        assert x.is_contiguous()
        y = torch.empty((M, D), dtype=bf16, device=x.device)
        stats = torch.empty((M, 2), dtype=f32, device=x.device) if need_stats else None
        fn, what = (self.lib.e4t_layernorm_fwd_f32, "e4t_layernorm_fwd_f32") if x.dtype == f32 else (self.lib.e4t_layernorm_fwd, "e4t_layernorm_fwd")
        self._timed("ln_fwd", 0.0, lambda: _C.check(fn(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), M, D, float(eps), _stream()), what),
                    (x.element_size() + 2.0) * M * D)
        return y, stats

    def layernorm_bwd(self, x, dy, gamma, stats, want_param_grads=False, add=None):
        M, D = x.shape
        assert dy.is_contiguous() and (add is None or (add.is_contiguous() and add.shape == x.shape and add.dtype == x.dtype))
        dx = torch.empty_like(x)
        self._timed("ln_bwd", 0.0, lambda: _C.check(self.lib.e4t_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(stats), _ptr(add), _ptr(dx), M, D,
                                                                           _stream()), "e4t_layernorm_bwd"), 2.0 * M * D * (4 if add is not None else 3))
        dgamma = dbeta = None
        if want_param_grads:
            dgamma, dbeta = torch.empty(D, dtype=f32, device=x.device), torch.empty(D, dtype=f32, device=x.device)
            ws = self.workspace(2 * self.lib.e4t_colreduce_splits(M) * D * 4, x.device)
            _C.check(self.lib.e4t_layernorm_param_grad(_ptr(x), _ptr(dy), _ptr(stats), M, D, _ptr(dgamma), _ptr(dbeta), 0, _ptr(ws), ws.numel(),
                                                       _stream()), "e4t_layernorm_param_grad")
        return dx, dgamma, dbeta

    def colsum(self, x, out=None, accumulate=False):
        """fp32 column sums of a (M, C) bf16 matrix (bias gradients); accumulate=True adds into `out`."""
        M, Cn = x.shape
        assert x.stride(1) == 1 and x.dtype == bf16
        if Cn % 8 or x.stride(0) % 8 or x.data_ptr() % 16:          # e.g. conv_out's 4 channels: not worth a kernel
            r = x.float().sum(0)
            return r if out is None else (out.add_(r) if accumulate else out.copy_(r))
        if out is None:
            out, accumulate = torch.empty(Cn, dtype=f32, device=x.device), False
        assert out.dtype == f32 and out.is_contiguous() and out.numel() == Cn
        ws = self.workspace(self.lib.e4t_colreduce_splits(M) * Cn * 4, x.device)
        _C.check(self.lib.e4t_colsum(_ptr(x), x.stride(0), M, Cn, _ptr(out), int(accumulate), _ptr(ws), ws.numel(), _stream()), "e4t_colsum")
        return out

    # ------------------------------------------------------------------ streaming ops
    def geglu_fwd(self, u):
        M, H2 = u.shape
        h = torch.empty((M, H2 // 2), dtype=bf16, device=u.device)
        self._timed("geglu_fwd", 0.0, lambda: _C.check(self.lib.e4t_geglu_fwd(_ptr(u), _ptr(h), M, H2 // 2, _stream()), "e4t_geglu_fwd"), 3.0 * M * H2)
        return h

    def geglu_bwd(self, u, dh):
        du = torch.empty_like(u)
        self._timed("geglu_bwd", 0.0, lambda: _C.check(self.lib.e4t_geglu_bwd(_ptr(u), _ptr(dh), _ptr(du), u.shape[0], u.shape[1] // 2, _stream()),
                                                       "e4t_geglu_bwd"), 5.0 * u.shape[0] * u.shape[1])
        return du

    def unary(self, x, op, dy=None):
        assert x.is_contiguous() and (dy is None or dy.is_contiguous())
        y = torch.empty_like(x)
        _C.check(self.lib.e4t_unary(_ptr(x), _ptr(dy), _ptr(y), x.numel(), op, _stream()), "e4t_unary")
        return y

    def add(self, a, b):
        assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
        y = torch.empty_like(a)
        _C.check(self.lib.e4t_add(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream()), "e4t_add")
        return y

    def transpose(self, x, pad_to=0):
        """bf16 [R, C] (row stride arbitrary) -> contiguous [C, max(R, pad_to)] (pad columns zero)."""
        _rowmajor(x, "transpose input")
        R, Cn = x.shape
        ldo = max(R, pad_to)
        out = torch.zeros((Cn, ldo), dtype=bf16, device=x.device) if ldo > R else torch.empty((Cn, ldo), dtype=bf16, device=x.device)
        _C.check(self.lib.e4t_transpose(_ptr(x), _ptr(out), 1, R, Cn, x.stride(0), ldo, 0, 0, _stream()), "e4t_transpose")
        return out

    def sumpool2(self, x, B, H, W):
        Cn = x.shape[-1]
        out = torch.empty((B * H * W, Cn), dtype=bf16, device=x.device)
        _C.check(self.lib.e4t_sumpool2(_ptr(x), _ptr(out), B, H, W, Cn, _stream()), "e4t_sumpool2")
        return out

    def spatial_mean(self, x, B, HW, out, coff):
        _C.check(self.lib.e4t_spatial_mean(_ptr(x), _ptr(out), B, HW, x.shape[-1], out.stride(0), coff, _stream()), "e4t_spatial_mean")

    def spatial_mean_bwd(self, g, base, B, HW, Cn, coff):
        dx = torch.empty((B * HW, Cn), dtype=bf16, device=g.device)
        _C.check(self.lib.e4t_spatial_mean_bwd(_ptr(g), _ptr(base), _ptr(dx), B, HW, Cn, g.stride(0), coff, _stream()), "e4t_spatial_mean_bwd")
        return dx

    def timestep_embedding(self, t, dim):
        t = t.to(torch.int64).contiguous()
        out = torch.empty((t.shape[0], dim), dtype=bf16, device=t.device)
        _C.check(self.lib.e4t_timestep_embedding(_ptr(t), _ptr(out), t.shape[0], dim, _stream()), "e4t_timestep_embedding")
        return out

    def clip_preprocess(self, pixels, S, P, Kpad):
        pixels = pixels.float().contiguous()
        B, _, Hin, Win = pixels.shape
        g = S // P
        out = torch.zeros((B * g * g, Kpad), dtype=bf16, device=pixels.device)
        _C.check(self.lib.e4t_clip_preprocess(_ptr(pixels), _ptr(out), B, Hin, Win, S, P, Kpad, _stream()), "e4t_clip_preprocess")
        return out

    def guided_step(self, pred, sample, coef, noise=None, cfg=True, pred_nhwc=False, out=None):
        """out = c_sample*sample + c_pred*(cfg ? u + g*(c-u) : pred) (+ c_noise*noise); coef = device fp32 [g, c_sample, c_pred, c_noise]"""
        B, Cn = sample.shape[0], sample.shape[1]
        HW = sample.numel() // (B * Cn)
        assert pred.dtype == f32 and sample.dtype == f32 and coef.dtype == f32 and coef.numel() == 4
        assert pred.is_contiguous() and sample.is_contiguous() and pred.numel() == sample.numel() * (2 if cfg else 1)
        assert noise is None or (noise.dtype == f32 and noise.is_contiguous() and noise.shape == sample.shape)
        if out is None:
            out = torch.empty_like(sample)
        _C.check(self.lib.e4t_guided_step(_ptr(pred), _ptr(sample), _ptr(noise), _ptr(out), _ptr(coef), B, Cn, HW, int(cfg), int(pred_nhwc), _stream()),
                 "e4t_guided_step")
        return out

    def image_prep(self, pool, table, B, S, out=None):
        """raw uint8 RGB images packed in `pool` + int64 [B,8] plan `table` (both on the device) -> fp32 [B,3,S,S]"""
        assert pool.dtype == torch.uint8 and table.dtype == torch.int64 and table.shape == (B, 8) and table.is_contiguous()
        if out is None:
            out = torch.empty((B, 3, S, S), dtype=torch.float32, device=pool.device)
        _C.check(self.lib.e4t_image_prep(_ptr(pool), _ptr(table), _ptr(out), B, S, _stream()), "e4t_image_prep")
        return out

    def adamw(self, p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
        self._timed("adamw", 0.0, lambda: _C.check(self.lib.e4t_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                                                                      grad_scale, _stream()), "e4t_adamw"), 28.0 * p.numel())

    def adamw_hyper(self, p, g, m, v, hyper, beta1, beta2, eps, wd):
        """the same update with {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale} read from the device tensor `hyper` (fp32 [4]): replayable"""
        self._timed("adamw", 0.0, lambda: _C.check(self.lib.e4t_adamw_hyper(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper), beta1, beta2, eps, wd,
                                                                            _stream()), "e4t_adamw_hyper"), 28.0 * p.numel())

    def adamw_rank(self, p, m, v, G, Z, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, hyper=None):
        """AdamW of the stack p [n, rows, cols] (fp32, with its moments) under the never-materialised gradient
        dW_i = grad_scale * G^T Z[:, i*cols:(i+1)*cols]   (G [K, rows], Z [K, n*cols], bf16).  hyper: device scalars as in adamw_hyper."""
        n, rows, cols = p.shape
        K = G.shape[0]
        assert p.is_contiguous() and m.is_contiguous() and v.is_contiguous() and G.dtype == bf16 and Z.dtype == bf16
        assert G.shape[1] == rows and Z.shape == (K, n * cols) and G.stride(1) == 1 and Z.stride(1) == 1
        self._timed("adamw", 2.0 * p.numel() * K, lambda: _C.check(self.lib.e4t_adamw_rank(
            _ptr(p), _ptr(m), _ptr(v), _ptr(G), _ptr(Z), n, rows, cols, K, G.stride(0), Z.stride(0), lr, beta1, beta2, eps, wd, int(step), grad_scale,
            _ptr(hyper), _stream()), "e4t_adamw_rank"), 24.0 * p.numel())

    def im2col_T(self, x, B, Hin, Win, Hout, Wout, mode):
        """bf16 [B*Hin*Win, C] -> [9*C, ld] with ld = B*Hout*Wout rounded up to 8 (zero padded): B operand of the wgrad GEMM"""
        assert x.is_contiguous()
        Cn = x.shape[1]
        ld = (B * Hout * Wout + 7) // 8 * 8
        out = torch.empty((9 * Cn, ld), dtype=bf16, device=x.device)
        _C.check(self.lib.e4t_im2col_T(_ptr(x), _ptr(out), B, Hin, Win, Cn, Hout, Wout, ld, mode, _stream()), "e4t_im2col_T")
        return out

    def im2col(self, x, B, Hin, Win, Hout, Wout, mode):
        """bf16 [B*Hin*Win, C] -> [B*Hout*Wout, 9*C] (tap-major columns): B operand of the TN weight-gradient GEMM"""
        assert x.is_contiguous() and x.shape[1] % 8 == 0
        Cn = x.shape[1]
        out = torch.empty((B * Hout * Wout, 9 * Cn), dtype=bf16, device=x.device)
        _C.check(self.lib.e4t_im2col(_ptr(x), _ptr(out), B, Hin, Win, Cn, Hout, Wout, mode, _stream()), "e4t_im2col")
        return out

    def softmax_rows_(self, x):
        """in-place softmax over the last dim of a bf16 matrix [..., L] with contiguous rows"""
        L = x.shape[-1]
        rows = x.numel() // L
        assert x.is_contiguous()
        _C.check(self.lib.e4t_softmax_rows(_ptr(x), rows, L, L, _stream()), "e4t_softmax_rows")
        return x

    def im2col3_rgb(self, pixels):
        pixels = pixels.float().contiguous()
        B, c, H, W = pixels.shape
        assert c == 3
        out = torch.empty((B * H * W, 32), dtype=bf16, device=pixels.device)
        _C.check(self.lib.e4t_im2col3_rgb(_ptr(pixels), _ptr(out), B, H, W, _stream()), "e4t_im2col3_rgb")
        return out

    def sumsq(self, g):
        nb = 1024
        part = torch.empty(nb, dtype=f32, device=g.device)
        _C.check(self.lib.e4t_sumsq_partial(_ptr(g), g.numel(), _ptr(part), nb, _stream()), "e4t_sumsq_partial")
        return part.sum()

    # ------------------------------------------------------------------ weight offsets
    def _pack_table(self, table: WOTable, device):
        sig = tuple((_ptr(e.W), _ptr(e.weff), _ptr(e.weffT), _ptr(e.dweff), _ptr(e.g_W),
                     _ptr(e.grads["g_v"]) if e.grads else 0) for e in table.entries)
        if table.dev is not None and table._sig == sig:
            return
        arr = (_C.WODesc * len(table.entries))()
        for d, e in zip(arr, table.entries):
            if e.params is not None:
                for k in ("v", "w1", "b1", "w2", "b2", "wc", "bc", "wr", "br"):
                    setattr(d, k, _ptr(e.params[k]))
                if e.vecs is None:
                    e.vecs = torch.empty(self.lib.e4t_wo_vecs_floats(e.row, e.col), dtype=f32, device=device)
                if e.partial is None:
                    e.partial = torch.empty(self.lib.e4t_wo_partial_floats(e.row, e.col), dtype=f32, device=device)
                d.vecs, d.partial = _ptr(e.vecs), _ptr(e.partial)
            d.W, d.weff, d.weffT, d.dweff, d.g_W = _ptr(e.W), _ptr(e.weff), _ptr(e.weffT), _ptr(e.dweff), _ptr(e.g_W)
            if e.grads is not None:
                for k in ("g_v", "g_w1", "g_b1", "g_w2", "g_b2", "g_wc", "g_bc", "g_wr", "g_br"):
                    setattr(d, k, _ptr(e.grads[k]))
            d.row, d.col, d.mode = e.row, e.col, e.mode
            d.ld_weff = e.weff.stride(0) if e.weff is not None else 0
            d.ld_weffT = e.weffT.stride(0) if e.weffT is not None else 0
            d.ld_dweff = e.dweff.stride(0) if e.dweff is not None else 0
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        table.dev = raw.to(device)
        table._sig = sig

    def wo_forward(self, table: WOTable):
        dev = table.entries[0].W.device
        self._pack_table(table, dev)
        _C.check(self.lib.e4t_wo_forward(_ptr(table.dev), len(table.entries), table.max_row, table.max_col, _stream()), "e4t_wo_forward")

    def wo_backward(self, table: WOTable, accumulate: bool):
        dev = table.entries[0].W.device
        self._pack_table(table, dev)
        _C.check(self.lib.e4t_wo_backward(_ptr(table.dev), len(table.entries), table.max_row, table.max_col, int(accumulate), _stream()), "e4t_wo_backward")

    def weight_prepare(self, table: WOTable):
        dev = table.entries[0].W.device
        self._pack_table(table, dev)
        _C.check(self.lib.e4t_weight_prepare(_ptr(table.dev), len(table.entries), table.max_row, table.max_col, _stream()), "e4t_weight_prepare")

    def probe_mfma(self, device):
        rows = torch.zeros((64, 16), dtype=f32, device=device)
        cols = torch.zeros((64, 16), dtype=f32, device=device)
        _C.check(self.lib.e4t_probe_mfma_layout(_ptr(rows), _ptr(cols), _stream()), "e4t_probe_mfma_layout")
        return rows, cols


_backend = None


def backend():
    """The active backend.  Created on first use; raises if the HIP library cannot be loaded."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(b):
    """TEST HOOK ONLY (tests/emu_backend.py).  The product never calls this."""
    global _backend
    old, _backend = _backend, b
    return old
