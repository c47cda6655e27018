"""E4TEncoder — drop-in for the reference's ``e4t/encoder.py`` (class E4TEncoder :78-168).

Same constructor keywords, same sub-module / parameter names (``clip_vision.*`` in open_clip's
VisionTransformer layout, ``unet_feature_embedder.{0,2}``, ``feature_linear``, ``first_linears.{i}``,
``final_linear``) so ``encoder.pt`` checkpoints load by key, same ``forward(x, unet_down_block_samples)``.

What runs instead of the reference's ops:
  * kornia bicubic resize + CLIP normalise + the 14x14/stride-14 patch-embed conv's im2col: ONE kernel
    (``clip_preprocess``) that writes patch rows, so ``conv1`` is a plain MFMA GEMM;
  * the open_clip ViT-H-14 tower ([3P], restated): LN / fused-QKV GEMM / flash attention / out-proj GEMM with
    fused residual / fc GEMM with fused exact GELU / proj GEMM with fused residual, 32 x;
  * the 13 spatial means (:147) in one pooling op over the NHWC maps;
  * the 129-iteration Python loop of tiny linears (:159-162, 258 launches forward in the reference):
    ``feature_linear(cat[h_i, u]) = W_fh h_i + (W_fu u + b_f)`` becomes ONE GEMM over all 129 slots with the
    u-term as a per-image row bias, and ``mean_i first_linears[i](z_i)`` becomes ONE batched GEMM whose batch
    dimension is reduced in the epilogue.  Its backward is 2 batched GEMMs writing dZ and all 129 weight
    gradients straight into their .grad storage.
"""
from __future__ import annotations



import torch
from torch import nn

from . import functional as Fn
from . import ops
from .utils import AttributeDict

f32 = torch.float32

VIT_ARCHS = {
    "ViT-H-14": dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0),
    "ViT-L-14": dict(image_size=224, patch_size=14, width=1024, layers=24, heads=16, mlp_ratio=4.0),
    "ViT-tiny-test": dict(image_size=28, patch_size=14, width=128, layers=2, heads=2, mlp_ratio=4.0),
}


# ------------------------------------------------------------------------------------------------
# [3P] open_clip VisionTransformer (proj=None, output_tokens=True) on the HIP kernels
# ------------------------------------------------------------------------------------------------
class _MHA(nn.Module):
    """Parameter container with nn.MultiheadAttention's names: in_proj_weight, in_proj_bias, out_proj.{weight,bias}."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _Mlp(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = nn.Linear(width, hidden)
        self.gelu = nn.GELU()
        self.c_proj = nn.Linear(hidden, width)


class _ResBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio):
        super().__init__()
        self.heads = heads
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _MHA(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = _Mlp(width, int(width * mlp_ratio))
        self._pq, self._po = Fn.PreparedLinear(self.attn.in_proj_weight), Fn.PreparedLinear(self.attn.out_proj.weight)
        self._pf, self._pp = Fn.PreparedLinear(self.mlp.c_fc.weight), Fn.PreparedLinear(self.mlp.c_proj.weight)

    def forward(self, h, B, T):
        w = h.shape[1]
        n = Fn.layer_norm(h, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        qkv = Fn.linear(n, self.attn.in_proj_weight, self.attn.in_proj_bias, self._pq)
        dh = w // self.heads
        o = Fn.attention(qkv, None, B, self.heads, T, T, dh, dh ** -0.5)
        f32s = h.dtype == torch.float32 and ops.ACT != torch.float32      # fp32 residual stream (frozen tower): h = h + proj(.) stays fp32
        h = Fn.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, self._po, residual=h, out_f32=f32s)
        n = Fn.layer_norm(h, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        if torch.is_grad_enabled() and (n.requires_grad or self.mlp.c_fc.weight.requires_grad):
            u = Fn.UnaryFn.apply(Fn.linear(n, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self._pf), 2)   # unfused GELU keeps a backward
        else:
            # frozen tower: GELU in the GEMM epilogue.  (B x 257 token rows: the library runs the first B x 257 - r rows as full tiles and
            # the r = M % 128 tail rows at the end of the same launch — gemm.hip plan_gemm_tail; round 4's row-panel variant of this call,
            # 16 panels of 256 patch rows + a class-token GEMM, measured slower and is gone)
            u = Fn.linear(n, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self._pf, gelu=True)
        return Fn.linear(u, self.mlp.c_proj.weight, self.mlp.c_proj.bias, self._pp, residual=h, out_f32=f32s)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(width, heads, mlp_ratio) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0):
        super().__init__()
        self.image_size, self.patch_size, self.width = image_size, patch_size, width
        self.grid = image_size // patch_size
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = nn.LayerNorm(width)
        self.proj = None
        self.output_tokens = True
        self.tokens_after_ln_post = False    # open_clip >= 2.20 behaviour when True
        self.f32_residual = True             # (bf16 stream in the frozen tower: e_hat error 1.0e-2 instead of 4.7e-3, time equal; profiles/r03_ab/r03e_*)
        self._pc = Fn.PreparedLinear(self.conv1.weight)

    def forward(self, pixels):
        """pixels: (B,3,H,W) fp32 in [-1,1] (the E4T encoder's input, pre-resize).  Returns (pooled, tokens)."""
        be = ops.backend()
        B = pixels.shape[0]
        P, g, w = self.patch_size, self.grid, self.width
        kpad = (3 * P * P + 7) // 8 * 8
        patches = be.clip_preprocess(pixels, self.image_size, P, kpad)
        if patches.dtype != ops.ACT:
            patches = patches.to(ops.ACT)
        x = Fn.linear(patches, self.conv1.weight, None, self._pc)                       # [B*g*g, w]
        pos = self.positional_embedding.to(x.dtype)
        cls = (self.class_embedding.to(x.dtype) + pos[0])[None, None, :].expand(B, 1, w)
        h = torch.cat([cls, x.view(B, g * g, w) + pos[1:][None]], dim=1).reshape(B * (g * g + 1), w).contiguous()
        T = g * g + 1
        h = Fn.layer_norm(h, self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps)
        # The residual stream of the frozen tower is kept in fp32: the out-proj / fc2 GEMMs add an fp32 residual and store fp32, the
        # LayerNorms read fp32 rows.  This is a deliberate precision IMPROVEMENT over the reference's --mixed_precision arithmetic, not a
        # restatement of it: open_clip's LayerNorm casts its result back to the input dtype, so under torch.autocast the tower's stream
        # is bf16 there (oracle/e4t_oracle.py::_ViTLayerNorm; round-3 review).  It costs nothing measurable and puts the
        # predicted embedding closer to the fp32 oracle than the stock-autocast run is.  A trainable tower (--unfreeze_clip_vision)
        # keeps the bf16 stream — the reference's own arithmetic — which its backward kernels use.
        if self.f32_residual and not (torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())):
            h = h.float()
        for blk in self.transformer.resblocks:
            h = blk(h, B, T)
        if h.dtype != ops.ACT:
            w_ = h.shape[1]
            if self.tokens_after_ln_post:
                h = Fn.layer_norm(h, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps).view(B, T, w_)
                return h[:, 0], h[:, 1:]
            h = h.view(B, T, w_)
            pooled = Fn.layer_norm(h[:, 0].contiguous(), self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
            return pooled, h[:, 1:].to(ops.ACT)
        if self.tokens_after_ln_post:
            h = Fn.layer_norm(h, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps).view(B, T, w)
            return h[:, 0], h[:, 1:]
        h = h.view(B, T, w)
        pooled = Fn.layer_norm(h[:, 0].contiguous(), self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        return pooled, h[:, 1:]


# ------------------------------------------------------------------------------------------------
# the 129-slot head
# ------------------------------------------------------------------------------------------------
class _HeadFn(torch.autograd.Function):
    """ybar[b] = mean_i ( W_i (W_fh hs[b,i] + W_fu u[b] + b_f) + b_i )      (encoder.py:159-165)

    Gradients of feature_linear / first_linears are written by the kernels directly into the parameters'
    persistent .grad storage (accumulating), like the weight-offset banks; autograd only carries dhs, du."""

    @staticmethod
    def forward(ctx, hs, u, enc):
        be = ops.backend()
        B, n, w = hs.shape
        hs2 = hs.reshape(B * n, w)
        wf, wfT = enc._pf.get()
        Ws, WsT = enc._stack_prepared()
        c = be.gemm(u, wf[:, w:], bias=enc.feature_linear.bias, out_dtype=f32)                    # [B, w] fp32: W_fu u + b_f
        Z = be.gemm(hs2, wf[:, :w], rowbias=c, rows_per_batch=n)                                  # [B*n, w]
        Zb = Z.view(B, n, w).permute(1, 0, 2)                                                     # batch i: rows b, stride n*w
        ybar = be.gemm(Zb, Ws, reduce_batch=True, alpha=1.0 / n, out_dtype=f32)                   # [B, w] fp32
        ybar = ybar + enc._bias_stack().mean(0)
        ctx.enc = enc
        ctx.save_for_backward(hs2, u, Z)
        ctx.dims = (B, n, w)
        return ybar

    @staticmethod
    def backward(ctx, g):
        be = ops.backend()
        enc = ctx.enc
        hs2, u, Z = ctx.saved_tensors
        B, n, w = ctx.dims
        act = ops.ACT
        wf, wfT = enc._pf.get()
        Ws, WsT = enc._stack_prepared()
        gs = (g.float() / n)
        gb = gs.to(act).contiguous()
        # dZ[b, i, :] = gb[b] . W_i            (batched over i, A broadcast)
        dZ = torch.empty((B, n, w), dtype=act, device=g.device)
        be.gemm(gb.unsqueeze(0).expand(n, B, w), WsT, out=dZ.permute(1, 0, 2))
        # first_linears grads: dW_i = gb^T z_i (contraction over the B images, zero-padded to a 64-wide K tile); db_i = sum_b gs
        gW, gB, gWf, gbf = enc._grad_storage()
        # (129 x [1280 x 1280] fp32 = 845 MB of gradient: the first write after the trainer zeroed it overwrites instead of accumulating —
        # half the traffic of this launch, 0.65 -> ~0.35 ms per step; a micro-batch accumulation step or plain autograd use accumulates)
        fresh = getattr(enc, "_stack_grad_is_zero", False)
        # Data parallel: dW_i is a rank-B product of two SMALL factors (gb: B x w, z_i: B x w), so the sum over the ranks of the 845 MB
        # stack equals ONE product over all ranks' rows.  The trainer's exchange gathers the factors (5 MB per rank) instead of
        # all-reducing the stack; it answers None (local product, the stack goes through the all-reduce) outside a synchronising
        # step or when it is switched off.  Only on the first write: an accumulated stack holds earlier micro-batches' LOCAL sums.
        gbx, Zx = gb, Z.view(B, n * w)
        ex = enc.exchange_head_factors
        if ex is not None and fresh:
            got = ex(gbx, Zx)
            if got is not None:
                gbx, Zx = got                                                                     # [world * B, w], [world * B, n * w]
        # Round 6: the stack's gradient need not exist at all.  A trainer that runs the optimiser right after this backward takes the two
        # factors (take_head_factors) and its AdamW forms dW_i = gb^T z_i in registers (e4t_adamw_rank): no 845 MB write here, no read
        # back, nothing to clear.  Declined (None hook / False) whenever the stack must exist: accumulated micro-batches, a gradient clip
        # over the whole gradient, a stack that rides the all-reduce (local factors under data parallelism), plain autograd use.
        if fresh and getattr(enc, "debug_check_stack", False):      # the overwrite-on-first-write contract (trainer.zero_grad's invariant)
            assert float(gW.abs().max()) == 0.0, "first_linears gradient stack was written behind zero_grad's back"
        take = enc.take_head_factors
        if not (take is not None and fresh and (ex is None or gbx is not gb) and take(gbx, Zx)):
            enc._stack_grad_is_zero = False
            bp = (gbx.shape[0] + 7) // 8 * 8
            gbT = be.transpose(gbx, pad_to=bp)                                                    # [w, bp]
            ZT = be.transpose(Zx, pad_to=bp)                                                      # [n*w, bp]
            be.gemm(gbT.unsqueeze(0).expand(n, w, bp), ZT.view(n, w, bp), out=gW, accum=not fresh)
        gB.add_(gs.sum(0)[None, :])
        # feature_linear: Z = hs W_fh^T + rowbias(c),  c = u W_fu^T + b_f
        dZ2 = dZ.view(B * n, w)
        dc = torch.empty((B, w), dtype=f32, device=g.device)
        be.spatial_mean(dZ2, B, n, dc, 0)
        dc = dc * float(n)
        dcb = dc.to(act)
        dhs = be.gemm(dZ2, wfT[:w]).view(B, n, w) if ctx.needs_input_grad[0] else None
        du = be.gemm(dcb, wfT[w:]) if ctx.needs_input_grad[1] else None
        gWf[:, :w].add_(Fn._weight_grad(dZ2, hs2))
        gWf[:, w:].add_(Fn._weight_grad(dcb, u))
        gbf.add_(dc.sum(0))
        return dhs, du, None


class E4TEncoder(nn.Module):
    def __init__(self, word_embedding_dim=768, block_out_channels=(320, 640, 1280, 1280), arch="ViT-H-14",
                 version="laion2b_s32b_b79k", antialias=False, freeze_clip_vision=True, **kwargs):
        super().__init__()
        assert not antialias, "antialiased resize is not part of the reference's training path (encoder.py:131-139)"
        self.config = AttributeDict(word_embedding_dim=word_embedding_dim, block_out_channels=tuple(block_out_channels), arch=arch,
                                    version=version, antialias=antialias, freeze_clip_vision=freeze_clip_vision, **kwargs)
        vit_cfg = kwargs.get("vit_cfg") or VIT_ARCHS[arch]
        self.clip_vision = VisionTransformer(**vit_cfg)   # random init: there is no network for `version` weights here
        hid = vit_cfg["width"]
        if freeze_clip_vision:
            self.clip_vision.requires_grad_(False)
        boc = tuple(block_out_channels)
        feat = boc[0] + sum(2 * c for c in boc) + sum(boc[:-1]) + boc[-1]        # 10880 for SD (encoder.py:102)
        self.unet_feature_embedder = nn.Sequential(nn.Linear(feat, hid), nn.LeakyReLU(), nn.Linear(hid, hid))
        self.feature_linear = nn.Linear(2 * hid, hid)
        if arch == "ViT-H-14":
            n_odd_layers = 128 + 1
        else:
            n_odd_layers = kwargs.get("n_odd_layers", None)
            assert n_odd_layers is not None, "You must specify `n_odd_layers`!"
            n_odd_layers = int(n_odd_layers)
        self.first_linears = nn.ModuleList([nn.Linear(hid, hid) for _ in range(n_odd_layers)])
        self.act = nn.LeakyReLU()
        self.final_linear = nn.Linear(hid, word_embedding_dim)
        self.image_size = vit_cfg["image_size"]
        self.antialias = antialias
        self.register_buffer("mean", torch.tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)
        self._p0 = Fn.PreparedLinear(self.unet_feature_embedder[0].weight)
        self._p2 = Fn.PreparedLinear(self.unet_feature_embedder[2].weight)
        self._pf = Fn.PreparedLinear(self.feature_linear.weight)
        self._pl = Fn.PreparedLinear(self.final_linear.weight)
        self._wstack = self._bstack = None
        self._stack_key = None
        self._gW = self._gB = None
        self.on_backward_done = None      # trainer hook: every encoder gradient is final (starts the head's all-reduce)
        self.exchange_head_factors = None     # trainer hook (data parallel): (gb, Z) -> every rank's rows of both, or None (_HeadFn.backward)
        self.take_head_factors = None         # trainer hook: (gb, Z) -> True when its AdamW will form the stack's gradient from them (_HeadFn.backward)

    @property
    def dtype(self):
        return self.final_linear.weight.dtype

    # ---- stacked views of the 129 first_linears (weights live in ONE [n, hid, hid] buffer) ----------------
    def _ensure_stacked(self):
        w0 = self.first_linears[0].weight
        n, hid = len(self.first_linears), w0.shape[0]
        ok = self._wstack is not None and self._wstack.device == w0.device and all(
            l.weight.data_ptr() == self._wstack[i].data_ptr() for i, l in ((0, self.first_linears[0]), (n - 1, self.first_linears[-1])))
        if not ok:
            ws = torch.stack([l.weight.data for l in self.first_linears]).contiguous()
            bs = torch.stack([l.bias.data for l in self.first_linears]).contiguous()
            for i, l in enumerate(self.first_linears):
                l.weight.data, l.bias.data = ws[i], bs[i]
            self._wstack, self._bstack = ws, bs
            self._stack_key = None
            self._gW = self._gB = None

    def adopt_stacks(self, wstack, bstack, gW=None, gB=None):
        """Let a trainer hand in flat-buffer views as the stacked storage (weights already re-pointed)."""
        self._wstack, self._bstack, self._gW, self._gB = wstack, bstack, gW, gB
        self._stack_key = None

    def _bias_stack(self):
        self._ensure_stacked()
        return self._bstack

    def _stack_prepared(self):
        self._ensure_stacked()
        w0 = self.first_linears[0].weight
        key = (w0._version, self.first_linears[-1].weight._version, ops.weights_epoch(), self._wstack.data_ptr())
        if key != self._stack_key:
            n, hid, _ = self._wstack.shape
            # The compute copies and the 129-entry descriptor table are built ONCE per parameter storage and re-used after every
            # optimiser step: re-creating 129 descriptors in Python and re-uploading them cost ~0.9 ms of idle GPU per step
            # (tools/idle_report.py: the one wo_apply -> wo_apply gap), the re-cast itself is a single grouped launch.
            tab = getattr(self, "_stack_table", None)
            if tab is None or tab[0] != (self._wstack.data_ptr(), n, hid, ops.ACT, self._wstack.device):
                ws = torch.empty((n * hid, hid), dtype=ops.ACT, device=self._wstack.device)
                tmpT = torch.empty((hid, n * hid), dtype=ops.ACT, device=self._wstack.device)
                # one grouped launch: n plain-weight entries, each writing its [hid, hid] block and its transposed block
                ents = [ops.WOEntry(row=hid, col=hid, W=self._wstack[i], weff=ws[i * hid:(i + 1) * hid],
                                    weffT=tmpT[:, i * hid:(i + 1) * hid]) for i in range(n)]
                tab = self._stack_table = ((self._wstack.data_ptr(), n, hid, ops.ACT, self._wstack.device), ops.WOTable(ents), ws, tmpT)
            _, table, ws, tmpT = tab
            ops.backend().weight_prepare(table)
            self._wsT = tmpT.view(hid, n, hid).permute(1, 0, 2)      # [n, hid(in), hid(out)] view, row stride n*hid
            self._ws = ws.view(n, hid, hid)
            self._stack_key = key
        return self._ws, self._wsT

    def _grad_storage(self):
        """Persistent .grad buffers for first_linears (stacked) and feature_linear; zeroed when autograd cleared them."""
        self._ensure_stacked()
        fl = self.first_linears
        if self._gW is None:
            self._gW, self._gB = torch.zeros_like(self._wstack), torch.zeros_like(self._bstack)
        if fl[0].weight.grad is None or fl[0].weight.grad.data_ptr() != self._gW[0].data_ptr():
            if fl[0].weight.grad is None:
                self._gW.zero_(); self._gB.zero_()
                for i, l in enumerate(fl):
                    l.weight.grad, l.bias.grad = self._gW[i], self._gB[i]
            else:   # foreign .grad tensors: adopt them only if they form one stack, else re-stack once
                g0 = fl[0].weight.grad
                hid = g0.shape[0]
                if all(l.weight.grad is not None and l.weight.grad.data_ptr() == g0.data_ptr() + i * hid * hid * 4 for i, l in enumerate(fl)) \
                        and all(l.bias.grad is not None and l.bias.grad.data_ptr() == fl[0].bias.grad.data_ptr() + i * hid * 4 for i, l in enumerate(fl)):
                    self._gW = torch.as_strided(g0, (len(fl), hid, hid), (hid * hid, hid, 1))
                    self._gB = torch.as_strided(fl[0].bias.grad, (len(fl), hid), (hid, 1))
                else:
                    self._gW = torch.stack([l.weight.grad for l in fl]); self._gB = torch.stack([l.bias.grad for l in fl])
                    for i, l in enumerate(fl):
                        l.weight.grad, l.bias.grad = self._gW[i], self._gB[i]
        f = self.feature_linear
        if f.weight.grad is None:
            f.weight.grad = torch.zeros_like(f.weight.data)
        if f.bias.grad is None:
            f.bias.grad = torch.zeros_like(f.bias.data)
        return self._gW, self._gB, f.weight.grad, f.bias.grad

    # ---- forward ------------------------------------------------------------------------------------------
    def vision_is_frozen(self) -> bool:
        return not any(p.requires_grad for p in self.clip_vision.parameters())

    def encode_vision(self, x):
        """The CLIP-ViT half of forward(): depends on the image only, so a trainer may launch it early (on a side stream)
        and hand the result back through forward(vision=...)."""
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.vision_is_frozen()):
            return self.clip_vision(x)

    def forward(self, x, unet_down_block_samples: tuple, vision=None):
        """x: (B,3,H,W) image in [-1,1]; unet_down_block_samples: the 13 maps from UNet(..., return_encoder_outputs=True);
        vision: optional precomputed encode_vision(x)."""
        act = ops.ACT
        B = x.shape[0]
        maps = []
        for s in unet_down_block_samples:                 # NCHW-shaped views of NHWC storage -> the [B*HW, C] matrices
            m = s.permute(0, 2, 3, 1)
            m = m.reshape(-1, m.shape[-1])
            maps.append(m if m.dtype == act and m.is_contiguous() else m.to(act).contiguous())
        pooled = Fn.spatial_mean_cat(B, maps)                                              # [B, 10880] fp32
        cb = self.on_backward_done
        if cb is not None and torch.is_grad_enabled() and pooled.requires_grad:
            # the gradient w.r.t. the pooled UNet features is the last thing the encoder's backward produces (the ViT tower,
            # when trainable, is younger and therefore already done): every encoder gradient is final here
            pooled.register_hook(lambda g: (cb(), None)[1])
        e0, e2 = self.unet_feature_embedder[0], self.unet_feature_embedder[2]
        u = Fn.linear(pooled.to(act), e0.weight, e0.bias, self._p0)
        u = Fn.linear(Fn.leaky_relu(u), e2.weight, e2.bias, self._p2)                      # [B, hid]
        cls, tokens = vision if vision is not None else self.encode_vision(x)
        hs = torch.cat([cls[:, None], tokens[:, 1::2]], dim=1).contiguous()               # [B, n, hid]  (:155-156)
        assert hs.shape[1] == len(self.first_linears), (hs.shape, len(self.first_linears))
        ybar = _HeadFn.apply(hs, u, self)                                                  # [B, hid] fp32
        y = Fn.leaky_relu(ybar.to(act))
        return Fn.linear(y, self.final_linear.weight, self.final_linear.bias, self._pl, out_f32=True)
