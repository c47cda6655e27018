"""AutoencoderKL.encode on the HIP kernels ("next" row N1 of SURVEY.md §8f: 558 GMAC/image, inside every
pre-training step at pretrain_e4t.py:597-599).

Same parameters and key names as ``checkpoint_trees.VAEEncoder`` (= the diffusers AutoencoderKL encoder checkpoint
layout); forward-only and frozen, so no autograd: the kernels are called directly.
  conv_in (3 -> 128): a 27-wide im2col written by one kernel + one GEMM (K = 32) instead of padding RGB to 64 ch;
  ResBlocks: GroupNorm+SiLU kernels + implicit-GEMM 3x3 convs with the shortcut add in the epilogue;
  Downsample2D(padding=0): the asymmetric (0,1,0,1) pad is a conv gather mode (E4T_CONV_S2A), no F.pad copy;
  mid attention (1 head, 512 wide, 4096 tokens): scores = batched GEMM, in-place row softmax, P.V = batched GEMM.
"""
from __future__ import annotations

import torch

from . import _C, ops
from .checkpoint_trees import VAEDecoder as _VAEDecoderTree
from .checkpoint_trees import VAEEncoder as _VAEEncoderTree


class VAEEncoder(_VAEEncoderTree):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._cache = None

    # ---- one-time (frozen) weight preparation ------------------------------------------------------
    def _prepare(self):
        key = (self.quant_conv.weight.data_ptr(), self.quant_conv.weight._version)     # frozen: prepared once
        if self._cache is not None and self._cache["key"] == key:
            return self._cache
        be = ops.backend()
        act = ops.ACT
        c = {"key": key}

        def conv(m):
            wf, _ = be.conv_weight_prepare(m.weight, want_dgrad=False)
            return wf, m.bias.detach().float().contiguous()

        def lin(w, b):
            w2 = w.detach().reshape(w.shape[0], -1).float()
            return w2.to(act).contiguous(), b.detach().float().contiguous()

        enc = self.encoder
        w_in, _ = be.conv_weight_prepare(enc.conv_in.weight, Ipad=3, want_dgrad=False)      # [128][27]
        w_in32 = torch.zeros((w_in.shape[0], 32), dtype=act, device=w_in.device)
        w_in32[:, :27] = w_in
        c["conv_in"] = (w_in32, enc.conv_in.bias.detach().float().contiguous())

        def res(r):
            d = dict(n1=(r.norm1.weight.detach().float(), r.norm1.bias.detach().float()), c1=conv(r.conv1),
                     n2=(r.norm2.weight.detach().float(), r.norm2.bias.detach().float()), c2=conv(r.conv2))
            d["sc"] = lin(r.conv_shortcut.weight, r.conv_shortcut.bias) if r.conv_shortcut is not None else None
            return d

        c["down"] = [dict(res=[res(r) for r in blk.resnets], ds=conv(blk.downsamplers[0].conv) if blk.downsamplers is not None else None)
                     for blk in enc.down_blocks]
        mid = enc.mid_block
        at = mid.attentions[0]
        wqkv = torch.cat([at.query.weight, at.key.weight, at.value.weight], 0)
        bqkv = torch.cat([at.query.bias, at.key.bias, at.value.bias], 0)
        c["mid"] = dict(r0=res(mid.resnets[0]), r1=res(mid.resnets[1]), gn=(at.group_norm.weight.detach().float(), at.group_norm.bias.detach().float()),
                        qkv=lin(wqkv, bqkv), proj=lin(at.proj_attn.weight, at.proj_attn.bias))
        c["norm_out"] = (enc.conv_norm_out.weight.detach().float(), enc.conv_norm_out.bias.detach().float())
        c["conv_out"] = conv(enc.conv_out)
        c["quant"] = lin(self.quant_conv.weight, self.quant_conv.bias)
        self._cache = c
        return c

    @staticmethod
    def _res(be, x, B, H, W, p):
        h, _ = be.groupnorm_fwd(x, None, p["n1"][0], p["n1"][1], B, H * W, 32, 1e-6, True)
        h = be.conv3x3(h, p["c1"][0], B, H, W, H, W, _C.CONV_S1, bias=p["c1"][1], colstats=True)
        h, _ = be.groupnorm_fwd(h, None, p["n2"][0], p["n2"][1], B, H * W, 32, 1e-6, True)
        s = x if p["sc"] is None else be.gemm(x, p["sc"][0], bias=p["sc"][1])
        return be.conv3x3(h, p["c2"][0], B, H, W, H, W, _C.CONV_S1, bias=p["c2"][1], residual=s, colstats=True)

    @torch.no_grad()
    def moments(self, x):
        """x: (B,3,H,W) image in [-1,1]  ->  (mean, logvar) each (B,4,H/8,W/8) fp32"""
        be = ops.backend()
        c = self._prepare()
        B, _, H, W = x.shape
        a = be.im2col3_rgb(x)
        if a.dtype != ops.ACT:
            a = a.to(ops.ACT)
        h = be.gemm(a, c["conv_in"][0], bias=c["conv_in"][1], colstats=True)      # (every GroupNorm input below carries column statistics)
        for blk in c["down"]:
            for r in blk["res"]:
                h = self._res(be, h, B, H, W, r)
            if blk["ds"] is not None:
                Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
                h = be.conv3x3(h, blk["ds"][0], B, H, W, Ho, Wo, _C.CONV_S2A, bias=blk["ds"][1], colstats=True)
                H, W = Ho, Wo
        m = c["mid"]
        h = self._res(be, h, B, H, W, m["r0"])
        # single-head attention over H*W tokens, width C (diffusers 0.14 AttentionBlock)
        T, Cw = H * W, h.shape[1]
        n, _ = be.groupnorm_fwd(h, None, m["gn"][0], m["gn"][1], B, T, 32, 1e-6, False)
        qkv = be.gemm(n, m["qkv"][0], bias=m["qkv"][1]).view(B, T, 3 * Cw)
        q, k, v = qkv[:, :, :Cw], qkv[:, :, Cw:2 * Cw], qkv[:, :, 2 * Cw:]
        s = be.gemm(q, k, alpha=Cw ** -0.5)                                        # [B, T, T] scores
        be.softmax_rows_(s)
        vt = torch.stack([be.transpose(v[b]) for b in range(B)])                   # [B, C, T]
        o = be.gemm(s, vt).view(B * T, Cw)
        h = be.gemm(o, m["proj"][0], bias=m["proj"][1], residual=h, colstats=True)
        h = self._res(be, h, B, H, W, m["r1"])
        n, _ = be.groupnorm_fwd(h, None, c["norm_out"][0], c["norm_out"][1], B, T, 32, 1e-6, True)
        y = be.conv3x3(n, c["conv_out"][0], B, H, W, H, W, _C.CONV_S1, bias=c["conv_out"][1])      # [B*T, 8]
        z = be.gemm(y, c["quant"][0], bias=c["quant"][1], out_dtype=torch.float32)                # 1x1 quant_conv
        z = z.view(B, H, W, -1).permute(0, 3, 1, 2)
        mean, logvar = z.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    @torch.no_grad()
    def encode_sample(self, x, eps):
        mean, logvar = self.moments(x)
        return (mean + torch.exp(0.5 * logvar) * eps) * self.scaling_factor


def _attention_1head(be, h, B, T, gn, qkv, proj):
    """diffusers 0.14 AttentionBlock at one head: GN -> q|k|v (one GEMM) -> scores GEMM -> in-place row softmax -> P.V -> proj + x"""
    Cw = h.shape[1]
    n, _ = be.groupnorm_fwd(h, None, gn[0], gn[1], B, T, 32, 1e-6, False)
    x = be.gemm(n, qkv[0], bias=qkv[1]).view(B, T, 3 * Cw)
    q, k, v = x[:, :, :Cw], x[:, :, Cw:2 * Cw], x[:, :, 2 * Cw:]
    s = be.gemm(q, k, alpha=Cw ** -0.5)
    be.softmax_rows_(s)
    vt = torch.stack([be.transpose(v[b]) for b in range(B)])
    o = be.gemm(s, vt).view(B * T, Cw)
    return be.gemm(o, proj[0], bias=proj[1], residual=h, colstats=True)


class VAEDecoder(_VAEDecoderTree):
    """AutoencoderKL.decode on the HIP kernels (inference row N4: pipeline_stable_diffusion_e4t.py:226,237 ->
    StableDiffusionPipeline.decode_latents).  Same parameters / key names as ``checkpoint_trees.VAEDecoder``.
      post_quant_conv (1x1, 4 -> 4) with 1/scaling_factor folded in: one GEMM whose output is already the 64-channel
        zero-padded NHWC operand of conv_in;
      Upsample2D: nearest x2 is a gather mode of the 3x3 conv (E4T_CONV_UP2) — the 4x larger map is never written;
      conv_out (128 -> 3): output channels padded to 8, written as fp32 NHWC = the layout decode_latents returns."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._cache = None

    def _prepare(self):
        key = (self.post_quant_conv.weight.data_ptr(), self.post_quant_conv.weight._version)
        if self._cache is not None and self._cache["key"] == key:
            return self._cache
        be, act = ops.backend(), ops.ACT
        c = {"key": key}
        dev = self.post_quant_conv.weight.device

        def conv(m):
            wf, _ = be.conv_weight_prepare(m.weight, want_dgrad=False)
            return wf, m.bias.detach().float().contiguous()

        def lin(w, b):
            return w.detach().reshape(w.shape[0], -1).float().to(act).contiguous(), b.detach().float().contiguous()

        def res(r):
            d = dict(n1=(r.norm1.weight.detach().float(), r.norm1.bias.detach().float()), c1=conv(r.conv1),
                     n2=(r.norm2.weight.detach().float(), r.norm2.bias.detach().float()), c2=conv(r.conv2))
            d["sc"] = lin(r.conv_shortcut.weight, r.conv_shortcut.bias) if r.conv_shortcut is not None else None
            return d

        L = self.post_quant_conv.weight.shape[0]
        wpq = torch.zeros((64, 8), dtype=torch.float32, device=dev)
        wpq[:L, :L] = self.post_quant_conv.weight.detach().float().reshape(L, L) / self.scaling_factor
        bpq = torch.zeros(64, dtype=torch.float32, device=dev)
        bpq[:L] = self.post_quant_conv.bias.detach().float()
        c["pq"] = (wpq.to(act).contiguous(), bpq)
        dec = self.decoder
        c["conv_in"] = conv(dec.conv_in)                                   # Cin padded to 64 by conv_weight_prepare
        mid, at = dec.mid_block, dec.mid_block.attentions[0]
        c["mid"] = dict(r0=res(mid.resnets[0]), r1=res(mid.resnets[1]),
                        gn=(at.group_norm.weight.detach().float(), at.group_norm.bias.detach().float()),
                        qkv=lin(torch.cat([at.query.weight, at.key.weight, at.value.weight], 0), torch.cat([at.query.bias, at.key.bias, at.value.bias], 0)),
                        proj=lin(at.proj_attn.weight, at.proj_attn.bias))
        c["up"] = [dict(res=[res(r) for r in blk.resnets], us=conv(blk.upsamplers[0].conv) if blk.upsamplers is not None else None)
                   for blk in dec.up_blocks]
        c["norm_out"] = (dec.conv_norm_out.weight.detach().float(), dec.conv_norm_out.bias.detach().float())
        wo = dec.conv_out.weight.detach()
        O = wo.shape[0]
        wo8 = torch.zeros((8,) + tuple(wo.shape[1:]), dtype=wo.dtype, device=dev)
        wo8[:O] = wo
        bo8 = torch.zeros(8, dtype=torch.float32, device=dev)
        bo8[:O] = dec.conv_out.bias.detach().float()
        c["conv_out"] = (be.conv_weight_prepare(wo8, want_dgrad=False)[0], bo8, O)
        self._cache = c
        return c

    @torch.no_grad()
    def decode_nhwc(self, latents):
        """latents (B,4,h,w) (scaled, as the scheduler leaves them) -> fp32 [B, 8h, 8w, 3] image in ~[-1, 1]"""
        be, act = ops.backend(), ops.ACT
        c = self._prepare()
        B, L, H, W = latents.shape
        z = torch.zeros((B * H * W, 8), dtype=act, device=latents.device)
        z[:, :L] = latents.permute(0, 2, 3, 1).reshape(B * H * W, L)
        h = be.gemm(z, c["pq"][0], bias=c["pq"][1])                                           # [B*H*W, 64], channels >= 4 are zero
        h = be.conv3x3(h, c["conv_in"][0], B, H, W, H, W, _C.CONV_S1, bias=c["conv_in"][1], colstats=True)
        m = c["mid"]
        h = VAEEncoder._res(be, h, B, H, W, m["r0"])
        h = _attention_1head(be, h, B, H * W, m["gn"], m["qkv"], m["proj"])
        h = VAEEncoder._res(be, h, B, H, W, m["r1"])
        for blk in c["up"]:
            for r in blk["res"]:
                h = VAEEncoder._res(be, h, B, H, W, r)
            if blk["us"] is not None:
                h = be.conv3x3(h, blk["us"][0], B, H, W, 2 * H, 2 * W, _C.CONV_UP2, bias=blk["us"][1], colstats=True)
                H, W = 2 * H, 2 * W
        n, _ = be.groupnorm_fwd(h, None, c["norm_out"][0], c["norm_out"][1], B, H * W, 32, 1e-6, True)
        w8, b8, O = c["conv_out"]
        y = be.conv3x3(n, w8, B, H, W, H, W, _C.CONV_S1, bias=b8, out_dtype=torch.float32)   # [B*H*W, 8] fp32
        return y.view(B, H, W, 8)[..., :O]

    @torch.no_grad()
    def decode(self, latents):
        return self.decode_nhwc(latents).permute(0, 3, 1, 2)

    @torch.no_grad()
    def decode_latents(self, latents):
        return (self.decode_nhwc(latents) / 2 + 0.5).clamp(0, 1).contiguous()
