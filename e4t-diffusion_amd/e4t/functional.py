"""Autograd glue between torch's graph and the HIP kernels: one ``torch.autograd.Function`` per
kernel family.  Forward and backward both call ``ops.backend()`` (the C ABI); nothing here
computes on tensors with torch except trivial bias-gradient row sums.

Activations are NHWC matrices: a feature map is a 2-D tensor [B*H*W, C] plus its (B, H, W) carried
in Python; tokens of a transformer block are the same matrix (no NCHW<->NLC permutes exist).
"""
from __future__ import annotations


import torch

from . import _C, ops

f32 = torch.float32


def act_dtype():
    return ops.ACT


# ----------------------------------------------------------------------------------------------
# prepared (compute-dtype) copies of fp32 master weights
# ----------------------------------------------------------------------------------------------
class PreparedLinear:
    """bf16 [N,K] and transposed [K,N] copies of an fp32 nn.Linear / 1x1-conv weight.
    Re-cast only when the master weight changed (frozen weights: exactly once)."""

    def __init__(self, weight: torch.nn.Parameter, colstats: bool = False):
        self.weight = weight
        self.key = None
        self.w = self.wT = None
        self._table = None
        self.colstats = colstats       # the layer's output feeds a GroupNorm: let the GEMM epilogue leave column statistics behind

    def get(self):
        wt = self.weight
        # frozen weights are cast exactly once; trainable ones also watch the epoch our raw-pointer AdamW bumps
        key = (wt._version, ops.weights_epoch() if wt.requires_grad else -1, wt.data_ptr(), wt.device)
        if key != self.key:
            W = wt.detach().reshape(wt.shape[0], -1)
            N, K = W.shape
            Kp = (K + 7) // 8 * 8          # pad K so that rows stay 16-B aligned (e.g. ViT patch embed 588 -> 592)
            Np = (N + 7) // 8 * 8
            # A trainable weight is re-cast after every optimiser step: the compute copies and the kernel's descriptor table
            # are kept, so that is one launch — not two allocations plus a pageable host-to-device copy of the descriptor
            # (measured: ~190 us of idle GPU in front of each of the E4T head's five re-casts per step).  Safe to overwrite in
            # place: the previous step's forward / backward readers are ahead of the re-cast on the stream.
            if self.w is None or self.w.shape != (N, Kp) or self.w.device != W.device or self.w.dtype != act_dtype():
                self.w = torch.zeros((N, Kp), dtype=act_dtype(), device=W.device)
                self.wT = torch.zeros((K, Np), dtype=act_dtype(), device=W.device)
                self._table = None
            if K % 4 or N % 4:
                self.w[:, :K] = W.to(act_dtype()); self.wT[:, :N] = W.t().to(act_dtype())   # odd shapes (never on the hot path)
            else:
                Wf = W.float().contiguous()
                if self._table is None or self._table.entries[0].W.data_ptr() != Wf.data_ptr():
                    self._table = ops.WOTable([ops.WOEntry(row=K, col=N, W=Wf, weff=self.w, weffT=self.wT)])
                ops.backend().weight_prepare(self._table)
            self.key = key
        return self.w, self.wT


class PreparedConv:
    """bf16 [O][3][3][Ipad] (forward) and [I][3][3][Opad] (dgrad, taps flipped) copies of an OIHW fp32 weight."""

    def __init__(self, weight: torch.nn.Parameter):
        self.weight = weight
        self.key = None
        self.wf = self.wd = None

    def get(self):
        wt = self.weight
        # frozen weights are cast exactly once; trainable ones also watch the epoch our raw-pointer AdamW bumps
        key = (wt._version, ops.weights_epoch() if wt.requires_grad else -1, wt.data_ptr(), wt.device)
        if key != self.key:
            self.wf, self.wd = ops.backend().conv_weight_prepare(wt.detach())
            self.key = key
        return self.wf, self.wd


def _pad8(n):
    return (n + 7) // 8 * 8


def _weight_grad(dy, x, x2=None, out=None):
    """dW[N, K] = dy^T . x  (fp32): the TN GEMM contracts over the rows of both operands as they lie in memory (no transposes).
    out: fp32 [N, K] to ACCUMULATE into (a parameter's .grad) instead of returning a fresh tensor."""
    be = ops.backend()
    if x2 is None:
        if out is not None:
            return be.gemm_tn(dy, x, out=out, accum=True)
        return be.gemm_tn(dy, x, out_dtype=f32)
    k1 = x.shape[1]
    res = out if out is not None else torch.empty((dy.shape[1], k1 + x2.shape[1]), dtype=f32, device=dy.device)
    be.gemm_tn(dy, x, out=res[:, :k1], accum=out is not None)          # fused-concat source 1 -> columns [0, k1)
    be.gemm_tn(dy, x2, out=res[:, k1:], accum=out is not None)         # source 2 -> columns [k1, K)
    return res


# Gradient accumulation without autograd's AccumulateGrad: when a trainer has installed persistent fp32 .grad storage
# (E4TTrainer's flat gradient buffer, zeroed once per step) and opted in, weight / bias gradients are accumulated into it
# by the producing kernel (GEMM ACCUM epilogue, colsum accumulate) and the autograd function returns None for them —
# in tuning that removes ~900 elementwise adds and as many temporaries per step.  Off by default: code that relies on
# autograd hooks on the parameters (e.g. torch DDP) must see the gradients come out of backward().
_INPLACE_PARAM_GRADS = False


def set_inplace_param_grads(enabled: bool):
    global _INPLACE_PARAM_GRADS
    _INPLACE_PARAM_GRADS = bool(enabled)


def _grad_slot(p):
    if not _INPLACE_PARAM_GRADS or p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != f32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


# ----------------------------------------------------------------------------------------------
# Linear / 1x1 conv:  y = act(x W^T + b) + residual
# ----------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, x2, prep: PreparedLinear, gelu: bool, out_f32: bool):
        be = ops.backend()
        w, wT = prep.get()
        K = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
        wv = w[:, :K] if w.shape[1] != K else w
        y = be.gemm(x, wv, a2=x2, bias=bias, residual=residual, gelu=gelu, out_dtype=f32 if out_f32 else act_dtype(), colstats=prep.colstats)
        ctx.prep, ctx.gelu = prep, gelu
        ctx.bias_param = bias if (bias is not None and bias.requires_grad and bias.is_leaf) else None
        ctx.N = weight.shape[0]
        need_w = weight.requires_grad
        ctx.save_for_backward(x if need_w else None, x2 if need_w else None)
        ctx.k1 = x.shape[1]
        ctx.has = (bias is not None, residual is not None, x2 is not None)
        ctx.res_dtype = residual.dtype if residual is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.gelu:
            raise NotImplementedError("LinearFn: backward through the fused GELU epilogue (frozen ViT only)")
        be = ops.backend()
        x, x2 = ctx.saved_tensors
        has_bias, has_res, has_x2 = ctx.has
        _, wT = ctx.prep.get()
        N = ctx.N
        dyb = dy if dy.dtype == act_dtype() else dy.to(act_dtype())
        dyb = dyb.contiguous()
        dx = dw = db = dres = dx2 = None
        wTn = wT[:, :N] if wT.shape[1] != N else wT
        if ctx.needs_input_grad[0]:
            dx = be.gemm(dyb, wTn[: ctx.k1])
        if has_x2 and ctx.needs_input_grad[4]:
            dx2 = be.gemm(dyb, wTn[ctx.k1:])
        if ctx.needs_input_grad[1]:
            wp = ctx.prep.weight
            kw = wp[0].numel()                       # true K (x may carry zero pad columns, e.g. patch-embed 588 -> 592)
            kx = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
            g = _grad_slot(wp)
            if g is not None and kx == kw:
                _weight_grad(dyb, x, x2, out=g.view(N, kw))      # accumulates straight into the (flat-buffer) .grad
            else:
                dw = _weight_grad(dyb, x, x2)
                if dw.shape[1] != kw:
                    dw = dw[:, :kw].contiguous()
                dw = dw.reshape(wp.shape)
        if has_bias and ctx.needs_input_grad[2]:
            g = _grad_slot(ctx.bias_param) if ctx.bias_param is not None else None
            if g is not None:
                ops.backend().colsum(dyb, out=g, accumulate=True)
            else:
                db = ops.backend().colsum(dyb)
        if has_res and ctx.needs_input_grad[3]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)
        return dx, dw, db, dres, dx2, None, None, None


def linear(x, weight, bias, prep, residual=None, x2=None, gelu=False, out_f32=False):
    return LinearFn.apply(x, weight, bias, residual, x2, prep, gelu, out_f32)


# ----------------------------------------------------------------------------------------------
# Weight-offset modulated projections (cross_attention.py:506,516,518)
# ----------------------------------------------------------------------------------------------
class WOSlot:
    """One (possibly fused q|k|v) projection: views into its bank's W_eff / W_eff^T / dW_eff buffers."""

    def __init__(self, weff, weffT, dweff):
        self.weff, self.weffT, self.dweff = weff, weffT, dweff
        self.dweff_valid = False


class WOLinearFn(torch.autograd.Function):
    """y = x . W_eff^T with W_eff = W o (1 + WO()) prepared by the bank; the backward accumulates
    dW_eff into the bank's fp32 buffer (both UNet passes land in the same buffer) and leaves the
    weight-offset parameter gradients to the bank's own backward node (reached through `token`)."""

    @staticmethod
    def forward(ctx, x, token, slot: WOSlot):
        y = ops.backend().gemm(x, slot.weff)
        ctx.slot = slot
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        (x,) = ctx.saved_tensors
        slot = ctx.slot
        dy = dy.contiguous()
        dx = be.gemm(dy, slot.weffT) if ctx.needs_input_grad[0] else None
        be.gemm_tn(dy, x, out=slot.dweff, accum=slot.dweff_valid)        # dW_eff (+)= dy^T . x
        slot.dweff_valid = True
        # no gradient VALUE flows to the token: the edge alone orders the bank's backward node behind every consumer (the engine counts
        # dependencies by edges and runs WOBankFn.backward with an undefined grad).  Returning a zero tensor here made the engine sum
        # 96 one-element tensors per step — 94 `add` launches of torch glue (round-4 review, weak #8).
        return dx, None, None


class WOBankFn(torch.autograd.Function):
    """Autograd node of a weight-offset bank.  forward: one grouped launch evaluates every W_eff of the
    bank; backward (runs after every consumer of `token` has accumulated its dW_eff): one grouped
    launch turns the dW_eff buffers into the gradients of all nine parameters of every instance,
    written straight into the parameters' .grad storage."""

    @staticmethod
    def forward(ctx, root, bank):
        bank._run_forward()
        ctx.bank = bank
        ctx.set_materialize_grads(False)          # consumers hand back no gradient value (WOLinearFn.backward): backward(None) still runs
        tok = getattr(bank, "_token_value", None)
        if tok is None or tok.device != root.device:
            tok = bank._token_value = torch.zeros(1, dtype=f32, device=root.device)     # (one fill per bank lifetime, not per forward)
        return tok.view_as(tok)

    @staticmethod
    def backward(ctx, g):
        ctx.bank._run_backward()
        return None, None


# ----------------------------------------------------------------------------------------------
# 3x3 convolution (implicit GEMM) — ResnetBlock2D convs, Downsample2D, Upsample2D, conv_in/out
# ----------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rowbias, residual, prep: PreparedConv, geom, mode: int, out_f32: bool):
        B, Hin, Win, Hout, Wout = geom
        wf, _ = prep.get()
        # every bf16 3x3-conv output of the UNet feeds a GroupNorm: the epilogue leaves its column statistics behind
        y = ops.backend().conv3x3(x, wf, B, Hin, Win, Hout, Wout, mode, bias=bias, rowbias=rowbias, residual=residual,
                                  out_dtype=f32 if out_f32 else act_dtype(), colstats=not out_f32)
        ctx.prep, ctx.geom, ctx.mode = prep, geom, mode
        ctx.cin = x.shape[1]
        ctx.has = (bias is not None, rowbias is not None, residual is not None)
        ctx.bias_param = bias if (bias is not None and bias.requires_grad and bias.is_leaf) else None
        ctx.res_dtype = residual.dtype if residual is not None else None
        ctx.save_for_backward(x if weight.requires_grad else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        B, Hin, Win, Hout, Wout = ctx.geom
        has_bias, has_rb, has_res = ctx.has
        dyb = dy if dy.dtype == act_dtype() else dy.to(act_dtype())
        dyb = dyb.contiguous()
        cout = dyb.shape[1]
        dx = db = drb = dres = None
        if ctx.needs_input_grad[0]:
            _, wd = ctx.prep.get()
            opad = wd.shape[1] // 9
            if opad != cout:   # conv_out: Cout = 4 padded to 64 in the dgrad weight layout
                t = torch.zeros((dyb.shape[0], opad), dtype=dyb.dtype, device=dyb.device)
                t[:, :cout] = dyb
                dyb_p = t
            else:
                dyb_p = dyb
            if ctx.mode == _C.CONV_S1:
                dx = be.conv3x3(dyb_p, wd, B, Hout, Wout, Hin, Win, _C.CONV_S1)
            elif ctx.mode == _C.CONV_S2:
                dx = be.conv3x3(dyb_p, wd, B, Hout, Wout, Hin, Win, _C.CONV_S2T)
            else:  # UP2: dgrad at the upsampled resolution, then fold the 2x2 replicas
                up = be.conv3x3(dyb_p, wd, B, Hout, Wout, Hout, Wout, _C.CONV_S1)
                dx = be.sumpool2(up, B, Hin, Win)
            if dx.shape[1] != ctx.cin:
                dx = dx[:, : ctx.cin].contiguous()
        dw = None
        if ctx.needs_input_grad[1]:
            # dW[co][tap][ci] = sum_pixels dY[m][co] * X[src(m, tap)][ci]: im2col of X, then one split-K TN GEMM contracting
            # over the pixels (tuning mode only; pre-training freezes these weights)
            (x,) = ctx.saved_tensors
            wshape = ctx.prep.weight.shape
            if cout % 8 == 0 and x.shape[1] % 8 == 0:
                dwk = be.gemm_tn(dyb, be.im2col(x, B, Hin, Win, Hout, Wout, ctx.mode), out_dtype=f32)    # [Cout, 9 * Cx], no transposes
            else:       # conv_out (4 output channels): the transposed route pads the row count instead
                xcolT = be.im2col_T(x, B, Hin, Win, Hout, Wout, ctx.mode)
                dwk = be.gemm(be.transpose(dyb, pad_to=xcolT.shape[1]), xcolT, out_dtype=f32)
            dw = dwk.view(wshape[0], 9, ctx.cin)[:, :, : wshape[1]].permute(0, 2, 1).reshape(wshape).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            g = _grad_slot(ctx.bias_param) if ctx.bias_param is not None else None
            if g is not None:
                ops.backend().colsum(dyb, out=g, accumulate=True)
            else:
                db = ops.backend().colsum(dyb)
        if has_rb and ctx.needs_input_grad[3]:
            drb = torch.zeros((B, cout), dtype=f32, device=dy.device)
            be.spatial_mean(dyb, B, Hout * Wout, drb, 0)
            drb = drb * float(Hout * Wout)
        if has_res and ctx.needs_input_grad[4]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)
        return dx, dw, db, drb, dres, None, None, None, None


def conv3x3(x, weight, bias, prep, geom, mode=_C.CONV_S1, rowbias=None, residual=None, out_f32=False):
    return ConvFn.apply(x, weight, bias, rowbias, residual, prep, geom, mode, out_f32)


class TimeEmbProjAllFn(torch.autograd.Function):
    """time_emb_proj(silu(emb)) of EVERY ResBlock of the UNet in one GEMM ([3P] ResnetBlock2D: `temb = time_emb_proj(act(temb))`,
    22 Linear(1280 -> Cout) on the same (B, 1280) input; M = B rows make each a latency-bound launch of its own).
    Returns one fp32 (B, Cout_i) column-slice view per block of a single (B, sum Cout) matrix; the conv epilogue reads its
    slice through the row stride (`ldrb`).  Frozen weights only (pretrain): dX = (cat of the slices' grads) . Wcat."""

    @staticmethod
    def forward(ctx, temb_act, wcat, wcat_t, bcat, splits):
        y = ops.backend().gemm(temb_act, wcat, bias=bcat, out_dtype=f32)
        ctx.save_for_backward(wcat_t)
        ctx.splits, ctx.act = splits, temb_act.dtype
        return tuple(y[:, o:o + c] for o, c in splits)

    @staticmethod
    def backward(ctx, *grads):
        (wcat_t,) = ctx.saved_tensors
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros((ref.shape[0], c)) for g, (o, c) in zip(grads, ctx.splits)]
        dy = torch.cat(parts, dim=1).to(ctx.act)
        return ops.backend().gemm(dy, wcat_t), None, None, None, None


# ----------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------
class GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, B, HW, G, eps, silu):
        y, stats = ops.backend().groupnorm_fwd(x1, x2, gamma, beta, B, HW, G, eps, silu)
        ctx.save_for_backward(x1, x2, stats, gamma, beta)
        ctx.cfg = (B, HW, G, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, stats, gamma, beta = ctx.saved_tensors
        B, HW, G, silu = ctx.cfg
        wp = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dx1, dx2, dg, db = ops.backend().groupnorm_bwd(x1, x2, dy.contiguous(), stats, gamma, beta, None, B, HW, G, silu, want_param_grads=wp)
        return dx1, dx2, dg, db, None, None, None, None, None


def group_norm(x1, x2, gamma, beta, B, HW, G, eps, silu):
    return GroupNormFn.apply(x1, x2, gamma, beta, B, HW, G, eps, silu)


class GroupNormSkipFn(torch.autograd.Function):
    """GroupNorm at the entry of a residual block: returns (act(GN(x1|x2)), x1, x2) where the last two are the inputs
    themselves, to be handed to the block's shortcut / residual.  In the backward the gradient that comes back through them
    is added inside the GroupNorm backward kernel instead of by separate elementwise adds (cf. LayerNormSkipFn)."""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, B, HW, G, eps, silu):
        y, stats = ops.backend().groupnorm_fwd(x1, x2, gamma, beta, B, HW, G, eps, silu)
        ctx.save_for_backward(x1, x2, stats, gamma, beta)
        ctx.cfg = (B, HW, G, silu)
        if x2 is None:
            return y, x1.view_as(x1)
        return y, x1.view_as(x1), x2.view_as(x2)

    @staticmethod
    def backward(ctx, dy, d1, d2=None):
        x1, x2, stats, gamma, beta = ctx.saved_tensors
        B, HW, G, silu = ctx.cfg
        wp = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        if dy is None:
            return d1, d2, None, None, None, None, None, None, None
        a1 = None if d1 is None else d1.to(x1.dtype).contiguous()
        a2 = None if d2 is None else d2.to(x2.dtype).contiguous()
        dx1, dx2, dg, db = ops.backend().groupnorm_bwd(x1, x2, dy.contiguous(), stats, gamma, beta, a1, B, HW, G, silu,
                                                       want_param_grads=wp, add2=a2)
        return dx1, dx2, dg, db, None, None, None, None, None


def group_norm_skip(x1, x2, gamma, beta, B, HW, G, eps, silu):
    """-> (act(GN(x1|x2)), x1, x2): use the returned x1 / x2 as the operands of the block's shortcut (see GroupNormSkipFn)."""
    if not (torch.is_grad_enabled() and (x1.requires_grad or (x2 is not None and x2.requires_grad))):
        return GroupNormFn.apply(x1, x2, gamma, beta, B, HW, G, eps, silu), x1, x2
    out = GroupNormSkipFn.apply(x1, x2, gamma, beta, B, HW, G, eps, silu)
    return (out[0], out[1], None) if x2 is None else out


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, stats = ops.backend().layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, stats, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma = ctx.saved_tensors
        wp = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dg, db = ops.backend().layernorm_bwd(x, dy.contiguous(), gamma, stats, want_param_grads=wp)
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


class LayerNormSkipFn(torch.autograd.Function):
    """Pre-LN residual block entry: returns (LN(x), x).  The second output is x itself, to be used as the block's residual
    operand; in the backward the gradient arriving through that residual is added inside the LayerNorm backward kernel
    instead of by a separate elementwise add of two (tokens x dim) tensors (attention.py:275-332 has 3 such per block)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, stats = ops.backend().layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, stats, gamma)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, stats, gamma = ctx.saved_tensors
        wp = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if dy is None:
            return dskip, None, None, None
        add = None if dskip is None else dskip.to(x.dtype).contiguous()
        dx, dg, db = ops.backend().layernorm_bwd(x, dy.contiguous(), gamma, stats, want_param_grads=wp, add=add)
        return dx, dg, db, None


def layer_norm_skip(x, gamma, beta, eps=1e-5):
    """-> (LN(x), x): use the second value as the residual operand of the block (see LayerNormSkipFn)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return LayerNormFn.apply(x, gamma, beta, eps), x
    return LayerNormSkipFn.apply(x, gamma, beta, eps)


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
class AttentionFn(torch.autograd.Function):
    """self: qkv = [B*T, 3d] (q | k | v column blocks), kv = None.  cross: qkv = q [B*T, d], kv = [B*S, 2d]."""

    @staticmethod
    def forward(ctx, qkv, kv, B, H, T, S, DH, scale, causal=False):
        d = H * DH
        if kv is None:
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        else:
            q, k, v = qkv, kv[:, :d], kv[:, d:]
        o, lse = ops.backend().attention_fwd(q, k, v, B, H, T, S, DH, scale, causal=causal)
        ctx.save_for_backward(qkv, kv, o, lse)
        ctx.cfg = (B, H, T, S, DH, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, kv, o, lse = ctx.saved_tensors
        B, H, T, S, DH, scale, causal = ctx.cfg
        d = H * DH
        dqkv = torch.empty_like(qkv)
        dkv = torch.empty_like(kv) if kv is not None else None
        if kv is None:
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        else:
            q, k, v = qkv, kv[:, :d], kv[:, d:]
            dq, dk, dv = dqkv, dkv[:, :d], dkv[:, d:]
        ops.backend().attention_bwd(q, k, v, o, do.contiguous(), lse, dq, dk, dv, B, H, T, S, DH, scale, causal=causal)
        return dqkv, dkv, None, None, None, None, None, None, None


def attention(qkv, kv, B, H, T, S, DH, scale, causal=False):
    return AttentionFn.apply(qkv, kv, B, H, T, S, DH, scale, causal)


# ----------------------------------------------------------------------------------------------
# streaming ops
# ----------------------------------------------------------------------------------------------
class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return ops.backend().geglu_fwd(u)

    @staticmethod
    def backward(ctx, dh):
        (u,) = ctx.saved_tensors
        return ops.backend().geglu_bwd(u, dh.contiguous())


def geglu(u):
    return GegluFn.apply(u)


class UnaryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.save_for_backward(x)
        ctx.op = op
        return ops.backend().unary(x, op)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.backend().unary(x, ctx.op + 1, dy.contiguous()), None


def silu(x):
    return UnaryFn.apply(x, _C.OP_SILU)


def leaky_relu(x):
    return UnaryFn.apply(x, _C.OP_LRELU)


def gelu(x):
    return UnaryFn.apply(x, _C.OP_GELU)


def quick_gelu(x):
    """x * sigmoid(1.702 x): hidden_act of the SD-1.x CLIP text encoder"""
    return UnaryFn.apply(x, _C.OP_QGELU)


class SpatialMeanFn(torch.autograd.Function):
    """cat([m.mean(dim=(2,3)) for m in maps], -1) for NHWC maps given as [B*HW, C] matrices (encoder.py:147-148)."""

    @staticmethod
    def forward(ctx, B, *maps):
        be = ops.backend()
        total = sum(m.shape[1] for m in maps)
        out = torch.empty((B, total), dtype=f32, device=maps[0].device)
        off, meta = 0, []
        for m in maps:
            hw = m.shape[0] // B
            be.spatial_mean(m, B, hw, out, off)
            meta.append((hw, m.shape[1], off))
            off += m.shape[1]
        ctx.meta, ctx.B = meta, B
        return out

    @staticmethod
    def backward(ctx, g):
        be = ops.backend()
        g = g.contiguous().float()
        outs = []
        for i, (hw, c, off) in enumerate(ctx.meta):
            outs.append(be.spatial_mean_bwd(g, None, ctx.B, hw, c, off) if ctx.needs_input_grad[1 + i] else None)
        return (None, *outs)


def spatial_mean_cat(B, maps):
    return SpatialMeanFn.apply(B, *maps)
