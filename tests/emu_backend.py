"""TEST-ONLY double of e4t.ops.HipBackend: every op restated with plain torch (fp32 math, outputs
rounded to the dtype the HIP kernel stores).  Two uses:
  * on the GPU box: per-op reference the HIP kernels are compared against (tests/test_kernels_gpu.py);
  * here (no GPU): swapped in through e4t.ops.set_backend() so the host-side graph logic (module
    wiring, hand-written backward orchestration, weight-offset banks) can be validated against the
    fp32 oracle on CPU.
The product never imports this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

bf16, f32 = torch.bfloat16, torch.float32
CONV_S1, CONV_S2, CONV_UP2, CONV_S2T = 1, 2, 3, 4
OP_SILU, OP_SILU_BWD, OP_GELU, OP_GELU_BWD, OP_LRELU, OP_LRELU_BWD, OP_QGELU, OP_QGELU_BWD = range(8)


def _dsilu(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def _dgelu(x):
    cdf = 0.5 * (1 + torch.erf(x / math.sqrt(2)))
    pdf = torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    return cdf + x * pdf


class EmuBackend:
    name = "emu"

    def __init__(self, round_bf16=True):
        # round_bf16=False keeps "bf16" tensors' VALUES in full fp32 precision (stored as fp32) —
        # used by host-logic tests that want to compare against the fp32 oracle tightly.
        self.round = round_bf16

    def _act(self, t):
        """what a kernel that stores bf16 would return"""
        return t.to(bf16) if self.round else t.float()

    # ------------------------------------------------------------------ GEMM
    def gemm(self, a, b, *, a2=None, bias=None, residual=None, rowbias=None, rows_per_batch=0, out=None,
             out_dtype=bf16, gelu=False, accum=False, alpha=1.0, reduce_batch=False, tile=0, splitk=0, colstats=False, panels=None):
        if panels is not None:      # e4t_gemm_desc.panel_*: only the listed row panels are computed / written
            pr, ps, po, pc = panels
            idx = (torch.arange(pc, device=a.device)[:, None] * ps + po + torch.arange(pr, device=a.device)[None, :]).reshape(-1)
            y = self.gemm(a[idx], b, bias=bias, residual=None if residual is None else residual[idx], out_dtype=out.dtype, gelu=gelu, alpha=alpha)
            out[idx] = y.to(out.dtype)
            return out
        A = a.float()
        if a2 is not None:
            A = torch.cat([A, a2.float()], dim=-1)
        Bm = b.float()
        y = alpha * (A @ Bm.transpose(-1, -2))
        if bias is not None:
            y = y + (bias.float()[:, None, :] if (bias.dim() == 2 and y.dim() == 3) else bias.float())
        if reduce_batch:
            y = y.sum(0)
        if rowbias is not None:
            y = y + rowbias.float().repeat_interleave(rows_per_batch, dim=0)
        if gelu:
            y = F.gelu(y)
        if residual is not None:
            y = y + residual.float()
        want = out.dtype if out is not None else out_dtype
        if accum:
            y = y + out.float()
        y = y.to(want) if (want == f32 or self.round) else y
        if out is not None:
            out.copy_(y)
            y = out
        self._attach_colstats(y, colstats)
        return y

    @staticmethod
    def _attach_colstats(y, wanted):
        """what the kernels' epilogue leaves behind: (sum, sum of squares) of every column per 32-row block"""
        if wanted and y.dim() == 2 and y.shape[0] % 32 == 0:
            blk = y.detach().float().reshape(y.shape[0] // 32, 32, y.shape[1])
            y._e4t_colstats = torch.stack([blk.sum(1), (blk * blk).sum(1)], dim=-1).contiguous()

    # ------------------------------------------------------------------ conv
    def conv_weight_prepare(self, w_oihw, Ipad=None, Opad=None, want_fwd=True, want_dgrad=True):
        O, I = w_oihw.shape[:2]
        Ipad = Ipad or (I + 63) // 64 * 64
        Opad = Opad or (O + 63) // 64 * 64
        w = w_oihw.detach().float()
        wf = wd = None
        if want_fwd:
            t = torch.zeros(O, 3, 3, Ipad, dtype=f32, device=w.device)
            t[..., :I] = w.permute(0, 2, 3, 1)
            wf = self._act(t.reshape(O, 9 * Ipad))
        if want_dgrad:
            t = torch.zeros(I, 3, 3, Opad, dtype=f32, device=w.device)
            t[..., :O] = w.flip(2, 3).permute(1, 2, 3, 0)
            wd = self._act(t.reshape(I, 9 * Opad))
        return wf, wd

    def gemm_tn(self, a, b, *, out=None, out_dtype=f32, accum=False, alpha=1.0, splitk=0):
        y = alpha * (a.float().t() @ b.float())
        if out is None:
            return y if out_dtype == f32 else self._act(y)
        out.copy_((out.float() + y if accum else y).to(out.dtype))
        return out

    def conv3x3(self, x, w, B, Hin, Win, Hout, Wout, mode, *, colstats=False, bias=None, residual=None, rowbias=None, out=None,
                out_dtype=bf16, accum=False, tile=0, splitk=0):
        Cin, Cout = x.shape[-1], w.shape[0]
        xi = x.float().reshape(B, Hin, Win, Cin).permute(0, 3, 1, 2)
        wk = w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        if mode == CONV_S1:
            y = F.conv2d(xi, wk, padding=1)
        elif mode == CONV_S2:
            y = F.conv2d(xi, wk, stride=2, padding=1)
        elif mode == CONV_UP2:
            y = F.conv2d(F.interpolate(xi, scale_factor=2.0, mode="nearest"), wk, padding=1)
        elif mode == 5:  # CONV_S2A: stride 2, pad (0,1,0,1)
            y = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wk, stride=2)
        else:  # CONV_S2T: zero-stuffed x (to Hout x Wout) then stride-1 conv
            z = torch.zeros(B, Cin, Hout, Wout, dtype=f32, device=x.device)
            z[:, :, ::2, ::2] = xi
            y = F.conv2d(z, wk, padding=1)
        assert y.shape[2] == Hout and y.shape[3] == Wout, (y.shape, Hout, Wout)
        y = y.permute(0, 2, 3, 1).reshape(B * Hout * Wout, Cout)
        if bias is not None:
            y = y + bias.float()
        if rowbias is not None:
            y = y + rowbias.float().repeat_interleave(Hout * Wout, dim=0)
        if residual is not None:
            y = y + residual.float()
        want = out.dtype if out is not None else out_dtype
        if accum:
            y = y + out.float()
        y = y.to(want) if (want == f32 or self.round) else y
        if out is not None:
            out.copy_(y)
            y = out
        self._attach_colstats(y, colstats)
        return y

    # ------------------------------------------------------------------ attention
    @staticmethod
    def _heads(t, B, L, H, DH):
        return t[:, : H * DH].float().reshape(B, L, H, DH).permute(0, 2, 1, 3)

    @staticmethod
    def _causal(s, causal):
        if causal:
            T, S = s.shape[-2:]
            s = s.masked_fill(torch.arange(S, device=s.device)[None, :] > torch.arange(T, device=s.device)[:, None], float("-inf"))
        return s

    def attention_fwd(self, q, k, v, B, H, T, S, DH, scale, out=None, need_lse=True, causal=False):
        Q, K, V = self._heads(q, B, T, H, DH), self._heads(k, B, S, H, DH), self._heads(v, B, S, H, DH)
        s = self._causal((Q @ K.transpose(-1, -2)) * scale, causal)
        lse = torch.logsumexp(s, dim=-1) / math.log(2.0)      # log2 units, as the kernel stores it
        o = (torch.softmax(s, dim=-1) @ V).permute(0, 2, 1, 3).reshape(B * T, H * DH)
        o = self._act(o)
        if out is not None:
            out.copy_(o)
            o = out
        return o, (lse if need_lse else None)

    def attention_bwd(self, q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, scale, causal=False):
        Q, K, V = self._heads(q, B, T, H, DH), self._heads(k, B, S, H, DH), self._heads(v, B, S, H, DH)
        dO = self._heads(do, B, T, H, DH)
        O = self._heads(o, B, T, H, DH)
        s = self._causal((Q @ K.transpose(-1, -2)) * scale, causal)
        P = torch.exp2(s / math.log(2.0) - lse[..., None])
        dV = P.transpose(-1, -2) @ dO
        dP = dO @ V.transpose(-1, -2)
        delta = (dO * O).sum(-1, keepdim=True)
        dS = P * (dP - delta)
        dQ = dS @ K * scale
        dK = dS.transpose(-1, -2) @ Q * scale

        def put(dst, g, L):
            dst[:, : H * DH].copy_(g.permute(0, 2, 1, 3).reshape(B * L, H * DH).to(dst.dtype))
        put(dq, dQ, T); put(dk, dK, S); put(dv, dV, S)

    # ------------------------------------------------------------------ norms
    def groupnorm_fwd(self, x1, x2, gamma, beta, B, HW, G, eps, silu):
        x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1)
        Cn = x.shape[-1]
        xg = x.reshape(B, HW, G, Cn // G)
        mean = xg.mean(dim=(1, 3))
        var = xg.var(dim=(1, 3), unbiased=False)
        cs1, cs2 = getattr(x1, "_e4t_colstats", None), (getattr(x2, "_e4t_colstats", None) if x2 is not None else None)
        if cs1 is not None and (x2 is None or cs2 is not None) and HW % 32 == 0:
            # the product takes mean / var from the producers' column statistics: they must describe THIS tensor
            self.colstats_hits = getattr(self, "colstats_hits", 0) + 1
            cs = cs1 if cs2 is None else torch.cat([cs1, cs2], dim=1)
            t = cs.reshape(B, HW // 32, G, Cn // G, 2).sum(dim=(1, 3)) / float(HW * (Cn // G))
            torch.testing.assert_close(t[..., 0], mean, rtol=1e-3, atol=1e-4)
            torch.testing.assert_close(t[..., 1] - t[..., 0] ** 2, var, rtol=1e-2, atol=1e-4)
        rstd = torch.rsqrt(var + eps)
        xh = (xg - mean[:, None, :, None]) * rstd[:, None, :, None]
        y = xh.reshape(B * HW, Cn) * gamma.float() + beta.float()
        if silu:
            y = F.silu(y)
        return self._act(y), torch.stack([mean, rstd], dim=-1).contiguous()

    def groupnorm_bwd(self, x1, x2, dy, stats, gamma, beta, add, B, HW, G, silu, want_param_grads=False, add2=None):
        x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1)
        Cn = x.shape[-1]
        cpg = Cn // G
        mean, rstd = stats[..., 0], stats[..., 1]
        xh = ((x.reshape(B, HW, G, cpg) - mean[:, None, :, None]) * rstd[:, None, :, None]).reshape(B, HW, Cn)
        dz = dy.float().reshape(B, HW, Cn)
        if silu:
            dz = dz * _dsilu(xh * gamma.float() + beta.float())
        dzg = (dz * gamma.float()).reshape(B, HW, G, cpg)
        xhg = xh.reshape(B, HW, G, cpg)
        n = cpg * HW
        s1 = dzg.sum(dim=(1, 3)) / n
        s2 = (dzg * xhg).sum(dim=(1, 3)) / n
        dx = rstd[:, None, :, None] * (dzg - s1[:, None, :, None] - xhg * s2[:, None, :, None])
        dx = dx.reshape(B * HW, Cn)
        C1 = x1.shape[-1]
        if add is not None:
            dx[:, :C1] += add.float()
        if add2 is not None:
            dx[:, C1:] += add2.float()
        dx1 = self._act(dx[:, :C1].contiguous())
        dx2 = self._act(dx[:, C1:].contiguous()) if x2 is not None else None
        dgamma = dbeta = None
        if want_param_grads:
            dbeta = dz.sum(dim=(0, 1))
            dgamma = (dz * xh).sum(dim=(0, 1))
        return dx1, dx2, dgamma, dbeta

    def layernorm_fwd(self, x, gamma, beta, eps, need_stats=True):
        xf = x.float()
        mean = xf.mean(-1)
        rstd = torch.rsqrt(xf.var(-1, unbiased=False) + eps)
        y = (xf - mean[:, None]) * rstd[:, None] * gamma.float() + beta.float()
        return self._act(y), (torch.stack([mean, rstd], dim=-1).contiguous() if need_stats else None)

    def layernorm_bwd(self, x, dy, gamma, stats, want_param_grads=False, add=None):
        xh = (x.float() - stats[:, :1]) * stats[:, 1:]
        dg = dy.float() * gamma.float()
        dx = stats[:, 1:] * (dg - dg.mean(-1, keepdim=True) - xh * (dg * xh).mean(-1, keepdim=True))
        if add is not None:
            dx = dx + add.float()
        dgamma = dbeta = None
        if want_param_grads:
            dgamma, dbeta = (dy.float() * xh).sum(0), dy.float().sum(0)
        return self._act(dx), dgamma, dbeta

    def colsum(self, x, out=None, accumulate=False):
        r = x.float().sum(0)
        if out is None:
            return r
        out.copy_(out + r if accumulate else r)
        return out

    # ------------------------------------------------------------------ streaming ops
    def geglu_fwd(self, u):
        a, g = u.float().chunk(2, dim=-1)
        return self._act(a * F.gelu(g))

    def geglu_bwd(self, u, dh):
        a, g = u.float().chunk(2, dim=-1)
        d = dh.float()
        return self._act(torch.cat([d * F.gelu(g), d * a * _dgelu(g)], dim=-1))

    def unary(self, x, op, dy=None):
        xf = x.float()
        d = dy.float() if dy is not None else None
        if op == OP_SILU:
            y = F.silu(xf)
        elif op == OP_SILU_BWD:
            y = d * _dsilu(xf)
        elif op == OP_GELU:
            y = F.gelu(xf)
        elif op == OP_GELU_BWD:
            y = d * _dgelu(xf)
        elif op == OP_LRELU:
            y = F.leaky_relu(xf, 0.01)
        elif op == OP_LRELU_BWD:
            y = torch.where(xf > 0, d, 0.01 * d)
        elif op == OP_QGELU:
            y = xf * torch.sigmoid(1.702 * xf)
        else:
            sg = torch.sigmoid(1.702 * xf)
            y = d * sg * (1 + 1.702 * xf * (1 - sg))
        return self._act(y)

    def add(self, a, b):
        return self._act(a.float() + b.float())

    def transpose(self, x, pad_to=0):
        R, Cn = x.shape
        out = torch.zeros((Cn, max(R, pad_to)), dtype=x.dtype, device=x.device)
        out[:, :R] = x.t()
        return out

    def sumpool2(self, x, B, H, W):
        Cn = x.shape[-1]
        return self._act(x.float().reshape(B, H, 2, W, 2, Cn).sum(dim=(2, 4)).reshape(B * H * W, Cn))

    def spatial_mean(self, x, B, HW, out, coff):
        Cn = x.shape[-1]
        out[:, coff:coff + Cn] = x.float().reshape(B, HW, Cn).mean(1)

    def spatial_mean_bwd(self, g, base, B, HW, Cn, coff):
        dx = (g[:, coff:coff + Cn].float() / HW)[:, None, :].expand(B, HW, Cn).reshape(B * HW, Cn)
        if base is not None:
            dx = dx + base.float()
        return self._act(dx)

    def timestep_embedding(self, t, dim):
        half = dim // 2
        freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=f32, device=t.device) / half)
        ang = t.float()[:, None] * freq[None]
        return self._act(torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1))

    def guided_step(self, pred, sample, coef, noise=None, cfg=True, pred_nhwc=False, out=None):
        g, cs, cp, cn = coef.tolist()
        B, Cn = sample.shape[0], sample.shape[1]
        p = pred.reshape((2 if cfg else 1) * B, -1)
        if pred_nhwc:
            p = p.view(p.shape[0], -1, Cn).permute(0, 2, 1)
        p = p.reshape((-1,) + tuple(sample.shape[1:]))
        if cfg:
            u, c = p.chunk(2)
            p = u + g * (c - u)
        r = cs * sample + cp * p
        if noise is not None:
            r = r + cn * noise
        if out is not None:
            out.copy_(r)
            return out
        return r

    def image_prep(self, pool, table, B, S, out=None):
        import numpy as np
        import image_prep_oracle as ipo
        pool_np = pool.cpu().numpy()
        res = []
        for off, H, W, nH, nW, y0, x0, flip in table.cpu().tolist():
            img = pool_np[off:off + H * W * 3].reshape(H, W, 3)
            assert (nH, nW) == ipo.smallest_max_size_dims(H, W, S) or (nH, nW) == (H, W)
            res.append(torch.from_numpy(ipo.image_prep(img, S, y0, x0, bool(flip))))
        r = torch.stack(res).to(pool.device)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def clip_preprocess(self, pixels, S, P, Kpad):
        x = F.interpolate(pixels.float(), size=(S, S), mode="bicubic", align_corners=True)
        x = (x + 1.0) / 2.0
        mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=x.device)[None, :, None, None]
        std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=x.device)[None, :, None, None]
        x = (x - mean) / std
        B = x.shape[0]
        g = S // P
        pt = x.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * P * P)
        out = torch.zeros((B * g * g, Kpad), dtype=f32, device=x.device)
        out[:, : 3 * P * P] = pt
        return self._act(out)

    def adamw(self, p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
        gg = g * grad_scale
        p.mul_(1 - lr * wd)
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        p.addcdiv_(m, v.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)

    def adamw_rank(self, p, m, v, G, Z, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, hyper=None):
        n, rows, cols = p.shape
        if hyper is not None:
            lr, bc1, bc2s, grad_scale = (float(x) for x in hyper.tolist())
        else:
            bc1, bc2s = 1 - beta1 ** step, math.sqrt(1 - beta2 ** step)
        g = torch.einsum("kr,knc->nrc", G.float(), Z.float().view(G.shape[0], n, cols)) * grad_scale
        p.mul_(1 - lr * wd)
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.addcdiv_(m, v.sqrt() / bc2s + eps, value=-lr / bc1)

    def im2col_T(self, x, B, Hin, Win, Hout, Wout, mode):
        Cn = x.shape[1]
        xi = x.float().reshape(B, Hin, Win, Cn).permute(0, 3, 1, 2)
        if mode == CONV_UP2:
            xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
        cols = F.unfold(xi, 3, padding=1, stride=2 if mode == CONV_S2 else 1)        # [B, C*9, Ho*Wo], index c*9 + tap
        assert cols.shape[2] == Hout * Wout
        cols = cols.view(B, Cn, 9, Hout * Wout).permute(2, 1, 0, 3).reshape(9 * Cn, B * Hout * Wout)   # (tap*C + c, m)
        ld = (B * Hout * Wout + 7) // 8 * 8
        out = torch.zeros((9 * Cn, ld), dtype=f32, device=x.device)
        out[:, : B * Hout * Wout] = cols
        return self._act(out)

    def im2col(self, x, B, Hin, Win, Hout, Wout, mode):
        t = self.im2col_T(x, B, Hin, Win, Hout, Wout, mode)
        return t[:, : B * Hout * Wout].t().contiguous()

    def softmax_rows_(self, x):
        x.copy_(torch.softmax(x.float(), dim=-1).to(x.dtype))
        return x

    def im2col3_rgb(self, pixels):
        B, c, H, W = pixels.shape
        cols = F.unfold(pixels.float(), 3, padding=1)                      # [B, c*9, H*W] with index c*9 + (ky*3+kx)
        cols = cols.view(B, 3, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 27)   # -> (ky*3+kx)*3 + c
        out = torch.zeros((B * H * W, 32), dtype=f32, device=pixels.device)
        out[:, :27] = cols
        return self._act(out)

    def sumsq(self, g):
        return (g.float() ** 2).sum()

    # ------------------------------------------------------------------ weight offsets
    @staticmethod
    def _wo_out(e):
        p = e.params
        v = p["v"].float()
        vx = p["w1"].float().reshape(-1) * v + p["b1"].float()
        vy = p["w2"].float().reshape(-1) * v + p["b2"].float()
        a = p["wc"].float() @ vx
        b = p["wr"].float() @ vy
        s = p["wr"].float().sum(1)
        out = b[:, None] * a[None, :] + s[:, None] * p["bc"].float()[None, :] + p["br"].float()[:, None]
        return out, (vx, vy, a, b, s)

    def wo_forward(self, table):
        for e in table.entries:
            W = e.W.float()
            if e.params is not None:
                out, _ = self._wo_out(e)
                W = out if (getattr(e, "mode", 0) & 2) else W * (1 + out)
            if e.weff is not None:
                e.weff[:, : e.row].copy_(W.to(e.weff.dtype))
            if e.weffT is not None:
                e.weffT[:, : e.col].copy_(W.t().to(e.weffT.dtype))

    weight_prepare = wo_forward

    def wo_backward(self, table, accumulate):
        for e in table.entries:
            p = e.params
            out, (vx, vy, a, b, s) = self._wo_out(e)
            dWe = e.dweff[:, : e.row].float()
            G = dWe * e.W.float()                       # (col, row)
            g = {}
            g["g_br"] = G.sum(1)
            db = G @ a
            ds = G @ p["bc"].float()
            g["g_bc"] = G.t() @ s
            da = G.t() @ b
            g["g_wc"] = da[:, None] * vx[None, :]
            dvx = p["wc"].float().t() @ da
            g["g_wr"] = db[:, None] * vy[None, :] + ds[:, None]
            dvy = p["wr"].float().t() @ db
            v = p["v"].float()
            g["g_w1"] = (dvx * v).reshape(p["w1"].shape)
            g["g_b1"] = dvx
            g["g_w2"] = (dvy * v).reshape(p["w2"].shape)
            g["g_b2"] = dvy
            g["g_v"] = ((dvx * p["w1"].float().reshape(-1)).sum() + (dvy * p["w2"].float().reshape(-1)).sum()).reshape(1)
            for k, val in g.items():
                dst = e.grads[k]
                if accumulate:
                    dst.add_(val.reshape(dst.shape))
                else:
                    dst.copy_(val.reshape(dst.shape))
            if e.g_W is not None:
                gw = dWe * (1 + out)
                if accumulate:
                    e.g_W.add_(gw)
                else:
                    e.g_W.copy_(gw)
