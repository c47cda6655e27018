"""CPU fp32 ORACLE for the E4T training hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file.  The product package (``e4t-diffusion_amd/``) never does: it fails loudly when the HIP
extension is missing instead of falling back to anything in here.

What this file is: a plain-PyTorch (CPU, fp32) restatement of the algorithm the reference runs on
the path BASELINE.json's ``north_star`` names, with the reference's parameter names so one seeded
state dict loads into both the oracle and the native modules.  Every class cites the reference
file:line it follows (paths relative to /root/reference).  Third-party leaves that are NOT under
/root/reference are restated from their published semantics (SURVEY.md §8a rows a6, a7, a9, a10,
a11): diffusers==0.14.0 (requirements.txt:1) ResnetBlock2D / Downsample2D / Upsample2D /
Timesteps / TimestepEmbedding / DDPMScheduler / AutoencoderKL encoder, open_clip
VisionTransformer (unpinned, requirements.txt:17; contemporaneous v2.16 semantics, tokens returned
before ``ln_post``), kornia resize (bicubic, align_corners=True, antialias=False) ==
``F.interpolate(mode="bicubic", align_corners=True)``.

PARITY PINNING.  The reference has no tests, golden vectors or fixtures for this path (SURVEY.md
§4, §8c), so the fixtures were produced by running the reference itself in the build container:
  * ``WeightOffsets`` is pinned bit-for-bit against e4t/weightoffsets.py (the one reference file that
    imports as is): tests/golden/weightoffsets_*.pt, generator tests/golden/make_golden.py.
  * ``CrossAttention`` (math and SDPA processors), the transformer / UNet blocks, ``UNet2DConditionModel``
    (SD-1 and SD-2 style configs: 13 encoder maps, sample, python-int timestep path, gradients of all
    96 x 9 weight-offset tensors) and ``E4TEncoder`` (output + every parameter gradient) are pinned against
    the reference's own e4t/models/*.py and e4t/encoder.py, executed UNMODIFIED on stand-ins for their
    third-party imports (tests/golden/shims, generator tests/golden/make_golden_models.py, fixtures
    tests/golden/reference_*.pt, replayed by tests/test_reference_golden.py at rtol 2e-5).
  * ``e4t_sample`` (the denoising loop) is pinned against the reference's StableDiffusionE4TPipeline.__call__
    (e4t/pipeline_stable_diffusion_e4t.py, run unmodified on a stand-in for the diffusers base pipeline, with the
    reference UNet, a stand-in E4T encoder — the real one hard-codes the full-size 10880 features —, the torch CLIP
    text twin and this file's DDIMScheduler): final latents with and without guidance.
  * ``e4t_losses`` + the optimiser step (the training-step glue, a11) are pinned against pretrain_e4t.py:561-584 and
    :597-654 themselves: those lines are read from the file and exec'd verbatim in a prepared namespace (reference UNet,
    stand-ins for accelerate / the diffusers scheduler and VAE objects); losses, model_pred, the domain embedding, every
    weight-offset gradient and the post-AdamW parameters are replayed (tests/golden/reference_step.pt); likewise the
    domain-tuning step, tuning_e4t.py:266-269 and :272-338 (every UNet parameter trainable, global gradient-norm clip):
    tests/golden/reference_tuning_step.pt.
  * tests/golden/reference_{unet,encoder}_wide.pt hold reference outputs at widths the native modules support (weights
    derived from the parameter names, not stored): tests compare the oracle AND the native modules with them directly.
  * Two third-party leaves are pinned against third-party code that IS installed (transformers 5.x): the CLIP text
    encoder against ``transformers.CLIPTextModel`` and the ViT against ``transformers.CLIPVisionModel`` (the same
    architecture as open_clip's, weights mapped by name; pooled = post-LN of the class token, tokens before ln_post)
    — tests/test_text_host_logic.py, tests/test_encoder_host_logic.py.
  * Still **parity unpinned**: the diffusers leaves (ResnetBlock2D / Down/Upsample2D / Timesteps /
    TimestepEmbedding / AutoencoderKL / schedulers) and kornia's resize (restated as the torch call it wraps) — the
    stand-ins use this file's restatements of them.  They are anchored on the
    reference's call sites and on the known answer it states (UNet encoder feature width 10880,
    e4t/models/unet_2d_condition.py:586).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------------
# a1  WeightOffsets  — e4t/weightoffsets.py:5-23
# --------------------------------------------------------------------------------------------
class WeightOffsets(nn.Module):
    """Literal restatement (two dense Linear passes) of e4t/weightoffsets.py:5-23."""

    def __init__(self, row_dim: int, column_dim: int):
        super().__init__()
        self.v = nn.Parameter(torch.ones(1))
        self.linear1 = nn.Linear(1, row_dim)
        self.linear2 = nn.Linear(1, column_dim)
        self.linear_column = nn.Linear(row_dim, row_dim)
        self.linear_row = nn.Linear(column_dim, column_dim)

    def forward(self) -> torch.Tensor:
        vx = self.linear1(self.v)                       # :15  (row,)
        vy = self.linear2(self.v)                       # :16  (col,)
        m = vx[:, None] * vy[None, :]                   # :18  (row, col)
        m = self.linear_column(m.T)                     # :20  (col, row)
        m = self.linear_row(m.T)                        # :22  (row, col)
        return m.T                                      # :23  (col, row)


def weight_offsets_closed_form(v, w1, b1, w2, b2, wc, bc, wr, br) -> torch.Tensor:
    """Closed form of WeightOffsets.forward (SURVEY.md §8a "WeightOffsets closed form").

    out[c, r] = a[r]*b[c] + bc[r]*s[c] + br[c]  with  a = Wc·vx, b = Wr·vy, s = rowsum(Wr).
    This is what the HIP kernel computes; tests check it against the literal module above.
    """
    vx = w1[:, 0] * v + b1
    vy = w2[:, 0] * v + b2
    a = wc @ vx
    b = wr @ vy
    s = wr.sum(dim=1)
    return b[:, None] * a[None, :] + s[:, None] * bc[None, :] + br[:, None]


# --------------------------------------------------------------------------------------------
# a2/a3  CrossAttention with weight offsets — e4t/models/cross_attention.py:22-99, 490-538
# --------------------------------------------------------------------------------------------
class CrossAttention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False,
                 upcast_attention=False):
        super().__init__()
        inner = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5                   # :59
        self.to_q = nn.Linear(query_dim, inner, bias=bias)              # :75-77
        self.to_k = nn.Linear(cross_attention_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])  # :83-85
        self.wo_q = WeightOffsets(query_dim, inner)                     # :97-99
        self.wo_k = WeightOffsets(cross_attention_dim, inner)
        self.wo_v = WeightOffsets(cross_attention_dim, inner)

    def forward(self, x, ctx=None):
        b, t, _ = x.shape
        q = F.linear(x, self.to_q.weight * (1 + self.wo_q()), self.to_q.bias)     # :506
        ctx = x if ctx is None else ctx                                           # :509-510
        k = F.linear(ctx, self.to_k.weight * (1 + self.wo_k()), self.to_k.bias)   # :516
        v = F.linear(ctx, self.to_v.weight * (1 + self.wo_v()), self.to_v.bias)   # :518
        h = self.heads
        dh = q.shape[-1] // h
        q = q.view(b, -1, h, dh).transpose(1, 2)
        k = k.view(b, -1, h, dh).transpose(1, 2)
        v = v.view(b, -1, h, dh).transpose(1, 2)
        # softmax(q k^T / sqrt(dh)) v  — :527-529 (SDPA) == :222-251,313-314 (math path)
        att = torch.softmax((q @ k.transpose(-1, -2)) * self.scale, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(b, -1, h * dh)
        return self.to_out[0](o)                                                  # :534-537


# --------------------------------------------------------------------------------------------
# a4  GEGLU / FeedForward / BasicTransformerBlock — e4t/models/attention.py:181-332,335-384,409-430
# --------------------------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)      # :419

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)         # :429
        return h * F.gelu(gate)                         # :430  (exact erf gelu)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])  # :369-376

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim, upcast_attention=False):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)                    # :228-236
        self.ff = FeedForward(dim)                                                 # :238
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)     # :241-250
        self.norm1 = nn.LayerNorm(dim)                                             # :259
        self.norm2 = nn.LayerNorm(dim)                                             # :268
        self.norm3 = nn.LayerNorm(dim)                                             # :273

    def forward(self, x, ctx):
        x = self.attn1(self.norm1(x)) + x               # :291-303
        x = self.attn2(self.norm2(x), ctx) + x          # :305-317
        x = self.ff(self.norm3(x)) + x                  # :320-330
        return x


# --------------------------------------------------------------------------------------------
# a5  Transformer2DModel (continuous branch) — e4t/models/transformer_2d.py:146-153,249-286
# --------------------------------------------------------------------------------------------
class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32,
                 use_linear_projection=False, upcast_attention=False):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)       # :149
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)                           # :151
            self.proj_out = nn.Linear(inner, in_channels)                          # :203
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)                        # :153
            self.proj_out = nn.Conv2d(inner, in_channels, 1)                       # :205
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim, upcast_attention)])

    def forward(self, x, ctx):
        b, c, hh, ww = x.shape
        res = x
        x = self.norm(x)                                                           # :253
        if not self.use_linear_projection:
            x = self.proj_in(x)
            x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)                      # :257
        else:
            x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
            x = self.proj_in(x)                                                    # :261
        for blk in self.transformer_blocks:
            x = blk(x, ctx)
        if not self.use_linear_projection:
            x = x.reshape(b, hh, ww, -1).permute(0, 3, 1, 2).contiguous()          # :280
            x = self.proj_out(x)
        else:
            x = self.proj_out(x)
            x = x.reshape(b, hh, ww, -1).permute(0, 3, 1, 2).contiguous()          # :284
        return x + res                                                             # :286


# --------------------------------------------------------------------------------------------
# a6/a7  [3P diffusers 0.14.0] ResnetBlock2D / Downsample2D / Upsample2D
#        constructed at e4t/models/unet_2d_blocks.py:481,522,760,804,881,900,1732,1774,1855,1872
# --------------------------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if temb is not None and self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))           # dropout p=0
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h                                    # output_scale_factor = 1.0


class Downsample2D(nn.Module):
    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:                           # VAE encoder variant: asymmetric pad
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, size=None):
        if size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=size, mode="nearest")
        return self.conv(x)


# --------------------------------------------------------------------------------------------
# a8  UNet blocks — e4t/models/unet_2d_blocks.py:454-551,727-934,1697-1901
# --------------------------------------------------------------------------------------------
class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, cin, cout, temb, layers, heads, ctx_dim, eps, groups, add_downsample, linproj, upcast):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])   # :758-771
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, cout // heads, cout, ctx_dim, groups, linproj, upcast) for _ in range(layers)])  # :773-785
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None          # :801-808

    def forward(self, x, temb, ctx):
        outs = ()
        for r, a in zip(self.resnets, self.attentions):  # :840-846
            x = a(r(x, temb), ctx)
            outs += (x,)
        if self.downsamplers is not None:                # :849-853
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, cin, cout, temb, layers, eps, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])   # :879-892
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x, temb):
        outs = ()
        for r in self.resnets:                           # :913-926
            x = r(x, temb)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, ch, temb, heads, ctx_dim, eps, groups, linproj, upcast):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups, eps) for _ in range(2)])   # :480-533
        self.attentions = nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, ctx_dim, groups, linproj, upcast)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)                     # :542
        x = self.attentions[0](x, ctx)                   # :544-548
        return self.resnets[1](x, temb)                  # :549


class CrossAttnUpBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, cin, cout, prev, temb, layers, heads, ctx_dim, eps, groups, add_upsample, linproj, upcast):
        super().__init__()
        res = []
        for i in range(layers):                          # :1728-1743
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, cout // heads, cout, ctx_dim, groups, linproj, upcast) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None                # :1773-1774

    def forward(self, x, skips, temb, ctx, upsample_size=None):
        for r, a in zip(self.resnets, self.attentions):
            s, skips = skips[-1], skips[:-1]             # :1793-1794
            x = torch.cat([x, s], dim=1)                 # :1795
            x = a(r(x, temb), ctx)                       # :1816-1821
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)     # :1825
        return x


class UpBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, cin, cout, prev, temb, layers, eps, groups, add_upsample):
        super().__init__()
        res = []
        for i in range(layers):                          # :1851-1866
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, skips, temb, upsample_size=None):
        for r in self.resnets:
            s, skips = skips[-1], skips[:-1]
            x = torch.cat([x, s], dim=1)                 # :1883
            x = r(x, temb)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


# --------------------------------------------------------------------------------------------
# a9  UNet2DConditionModel — e4t/models/unet_2d_condition.py:38-288 (ctor), 410-562 (forward)
# --------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos=True, freq_shift=0) -> torch.Tensor:
    """[3P diffusers 0.14] Timesteps(dim, flip_sin_to_cos, freq_shift) used at unet_2d_condition.py:122,461."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    ang = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


SD14_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32,
    norm_eps=1e-5, use_linear_projection=False, upcast_attention=False,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
)
SD21_UNET_CONFIG = dict(SD14_UNET_CONFIG, sample_size=96, cross_attention_dim=1024,
                        attention_head_dim=(5, 10, 20, 20), use_linear_projection=True, upcast_attention=True)   # stable-diffusion-2-1 @768


class UNet2DConditionModel(nn.Module):
    def __init__(self, sample_size=64, in_channels=4, out_channels=4,
                 down_block_types=SD14_UNET_CONFIG["down_block_types"],
                 up_block_types=SD14_UNET_CONFIG["up_block_types"],
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, cross_attention_dim=768,
                 attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5, use_linear_projection=False,
                 upcast_attention=False, flip_sin_to_cos=True, freq_shift=0, **unused):
        super().__init__()
        boc = tuple(block_out_channels)
        n = len(boc)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * n
        temb = boc[0] * 4
        self.cfg = dict(block_out_channels=boc, flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)                  # :106-108
        self.time_embedding = TimestepEmbedding(boc[0], temb)                        # :129-135
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, typ in enumerate(down_block_types):                                   # :169-194
            in_ch, out_ch = out_ch, boc[i]
            last = i == n - 1
            if typ == "CrossAttnDownBlock2D":
                blk = CrossAttnDownBlock2D(in_ch, out_ch, temb, layers_per_block, attention_head_dim[i],
                                           cross_attention_dim, norm_eps, norm_num_groups, not last,
                                           use_linear_projection, upcast_attention)
            else:
                blk = DownBlock2D(in_ch, out_ch, temb, layers_per_block, norm_eps, norm_num_groups, not last)
            self.down_blocks.append(blk)
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb, attention_head_dim[-1], cross_attention_dim,
                                                 norm_eps, norm_num_groups, use_linear_projection,
                                                 upcast_attention)                  # :197-211
        self.up_blocks = nn.ModuleList()
        rboc = list(reversed(boc))
        rheads = list(reversed(attention_head_dim))
        out_ch = rboc[0]
        for i, typ in enumerate(up_block_types):                                     # :236-271
            last = i == n - 1
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, n - 1)]
            if typ == "CrossAttnUpBlock2D":
                blk = CrossAttnUpBlock2D(in_ch, out_ch, prev, temb, layers_per_block + 1, rheads[i],
                                         cross_attention_dim, norm_eps, norm_num_groups, not last,
                                         use_linear_projection, upcast_attention)
            else:
                blk = UpBlock2D(in_ch, out_ch, prev, temb, layers_per_block + 1, norm_eps, norm_num_groups, not last)
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)     # :275-278
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)                # :285-287

    def forward(self, sample, timestep, encoder_hidden_states, return_encoder_outputs=False):
        t = timestep
        if not torch.is_tensor(t):                                                   # :446-455
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])                                                # :458
        emb = self.time_embedding(timestep_embedding(t, self.cfg["block_out_channels"][0],
                                                     self.cfg["flip_sin_to_cos"], self.cfg["freq_shift"]))  # :461-468
        x = self.conv_in(sample)                                                     # :481
        skips = (x,)
        for blk in self.down_blocks:                                                 # :484-496
            if blk.has_cross_attention:
                x, outs = blk(x, emb, encoder_hidden_states)
            else:
                x, outs = blk(x, emb)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)                            # :508-515
        if return_encoder_outputs:                                                   # :517-521
            return dict(down_block_samples=skips + (x,))
        for blk in self.up_blocks:                                                   # :527-551
            k = len(blk.resnets)
            s, skips = skips[-k:], skips[:-k]
            if blk.has_cross_attention:
                x = blk(x, s, emb, encoder_hidden_states)
            else:
                x = blk(x, s, emb)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))                             # :554-557
        return x


# --------------------------------------------------------------------------------------------
# a10  E4TEncoder — e4t/encoder.py:78-168 ; [3P open_clip] VisionTransformer restated
# --------------------------------------------------------------------------------------------
class _ViTMLP(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = nn.Linear(width, hidden)
        self.gelu = nn.GELU()
        self.c_proj = nn.Linear(hidden, width)

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class _ViTLayerNorm(nn.LayerNorm):
    """[3P] open_clip.transformer.LayerNorm: torch's LayerNorm with the result cast back to the input dtype.  In fp32 it IS
    nn.LayerNorm; under torch.autocast(bf16) — the reference's --mixed_precision run — it hands a bf16 tensor to `x + attn(...)`,
    so the tower's residual stream is bf16 there (a plain nn.LayerNorm would return fp32 and promote the stream to fp32, which
    understates the reference's own bf16 error in the calibration leg of tests/parity_step.py; round-3 review)."""

    def forward(self, x):
        return super().forward(x).to(x.dtype)


class _ViTBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = _ViTLayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads, batch_first=True)
        self.ln_2 = _ViTLayerNorm(width)
        self.mlp = _ViTMLP(width, int(width * mlp_ratio))

    def forward(self, x):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _ViTTransformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio):
        super().__init__()
        self.resblocks = nn.ModuleList([_ViTBlock(width, heads, mlp_ratio) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


VIT_H_14 = dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0)


class VisionTransformer(nn.Module):
    """[3P] open_clip VisionTransformer with proj=None, output_tokens=True (e4t/encoder.py:91-97)."""

    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0,
                 tokens_after_ln_post=False):
        super().__init__()
        self.grid = image_size // patch_size
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid ** 2 + 1, width))
        self.ln_pre = _ViTLayerNorm(width)
        self.transformer = _ViTTransformer(width, layers, heads, mlp_ratio)
        self.ln_post = _ViTLayerNorm(width)
        self.tokens_after_ln_post = tokens_after_ln_post

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        # open_clip casts both embeddings to the activations' dtype (`self.class_embedding.to(x.dtype) + zeros(..., dtype=x.dtype)`,
        # `x + self.positional_embedding.to(x.dtype)`): no-ops in fp32 — this oracle — but under torch.autocast(bf16), the calibration leg
        # of tests/parity_step.py, they are what keeps the residual stream in bf16.  Without them torch.cat([fp32 cls, bf16 x]) promotes
        # the stream to fp32 for all 32 blocks and the calibration understates the reference's own bf16 error (round 6: the README
        # recipe's trainable tower, which computes on open_clip's bf16 stream, sat at 0.92 of a bound calibrated on an fp32 stream).
        cls = self.class_embedding.to(x.dtype)[None, None, :].expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.transformer(self.ln_pre(x))
        if self.tokens_after_ln_post:
            x = self.ln_post(x)
            return x[:, 0], x[:, 1:]
        return self.ln_post(x[:, 0]), x[:, 1:]


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class E4TEncoder(nn.Module):
    def __init__(self, word_embedding_dim=768, block_out_channels=(320, 640, 1280, 1280), vit_cfg=None,
                 n_odd_layers=None, freeze_clip_vision=True, tokens_after_ln_post=False):
        super().__init__()
        vit_cfg = dict(VIT_H_14 if vit_cfg is None else vit_cfg)
        self.clip_vision = VisionTransformer(**vit_cfg, tokens_after_ln_post=tokens_after_ln_post)  # :91-96
        hid = vit_cfg["width"]
        if freeze_clip_vision:
            self.clip_vision.requires_grad_(False)                                  # :98-99
        boc = tuple(block_out_channels)
        # 13 maps for the 4-level / 2-layers-per-block UNet: conv_in, (2 res + 1 down) x3, 2 res, mid
        feat = boc[0] + sum(2 * c for c in boc) + sum(boc[:-1]) + boc[-1]
        self.unet_feature_embedder = nn.Sequential(nn.Linear(feat, hid), nn.LeakyReLU(), nn.Linear(hid, hid))  # :101-105
        self.feature_linear = nn.Linear(2 * hid, hid)                               # :106
        grid = vit_cfg["image_size"] // vit_cfg["patch_size"]
        if n_odd_layers is None:
            n_odd_layers = (grid * grid) // 2 + 1                                   # :111  (128 + 1 for ViT-H-14)
        self.first_linears = nn.ModuleList([nn.Linear(hid, hid) for _ in range(n_odd_layers)])  # :117-123
        self.act = nn.LeakyReLU()
        self.final_linear = nn.Linear(hid, word_embedding_dim)                      # :125
        self.image_size = vit_cfg["image_size"]
        self.register_buffer("mean", torch.tensor(CLIP_MEAN), persistent=False)     # :128-129
        self.register_buffer("std", torch.tensor(CLIP_STD), persistent=False)

    def preprocess(self, x):                                                        # :131-139
        x = F.interpolate(x, size=(self.image_size, self.image_size), mode="bicubic", align_corners=True)
        x = (x + 1.0) / 2.0
        return (x - self.mean[None, :, None, None]) / self.std[None, :, None, None]

    def forward(self, x, unet_down_block_samples):
        pooled = torch.cat([s.mean(dim=(2, 3)) for s in unet_down_block_samples], dim=-1)   # :147-148
        u = self.unet_feature_embedder(pooled)                                      # :149
        cls, tokens = self.clip_vision(self.preprocess(x))                          # :153-154
        hs = torch.cat([cls[:, None], tokens[:, 1::2]], dim=1)                      # :155-156
        ys = []
        for i in range(hs.shape[1]):                                                # :159-162
            z = self.feature_linear(torch.cat([hs[:, i], u], dim=-1))
            ys.append(self.first_linears[i](z))
        y = self.act(torch.stack(ys).mean(dim=0))                                   # :163-166
        return self.final_linear(y)                                                 # :168


# --------------------------------------------------------------------------------------------
# #8  CLIP text encoder with inputs_embeds — e4t/models/modeling_clip.py:9-82 ([3P] transformers CLIPTextModel)
# --------------------------------------------------------------------------------------------
CLIP_TEXT_L = dict(vocab=49409, width=768, layers=12, heads=12, mlp=3072, max_len=77, act="quick_gelu")
CLIP_TEXT_H = dict(vocab=49409, width=1024, layers=23, heads=16, mlp=4096, max_len=77, act="gelu")


class _TextLayer(nn.Module):
    def __init__(self, width, heads, mlp, act):
        super().__init__()
        self.heads, self.act = heads, act
        self.layer_norm1 = nn.LayerNorm(width)
        self.q_proj = nn.Linear(width, width)
        self.k_proj = nn.Linear(width, width)
        self.v_proj = nn.Linear(width, width)
        self.out_proj = nn.Linear(width, width)
        self.layer_norm2 = nn.LayerNorm(width)
        self.fc1 = nn.Linear(width, mlp)
        self.fc2 = nn.Linear(mlp, width)

    def forward(self, x, mask):
        b, s, w = x.shape
        h = self.layer_norm1(x)
        dh = w // self.heads
        q = self.q_proj(h).view(b, s, self.heads, dh).transpose(1, 2)
        k = self.k_proj(h).view(b, s, self.heads, dh).transpose(1, 2)
        v = self.v_proj(h).view(b, s, self.heads, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5 + mask, dim=-1)
        x = x + self.out_proj((att @ v).transpose(1, 2).reshape(b, s, w))
        h = self.fc1(self.layer_norm2(x))
        h = h * torch.sigmoid(1.702 * h) if self.act == "quick_gelu" else F.gelu(h)
        return x + self.fc2(h)


class CLIPTextModel(nn.Module):
    def __init__(self, vocab=49409, width=768, layers=12, heads=12, mlp=3072, max_len=77, act="quick_gelu"):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.position_embedding = nn.Embedding(max_len, width)
        self.layers = nn.ModuleList([_TextLayer(width, heads, mlp, act) for _ in range(layers)])
        self.final_layer_norm = nn.LayerNorm(width)

    def get_input_embeddings(self):
        return self.token_embedding

    def forward(self, input_ids=None, inputs_embeds=None):
        if inputs_embeds is None:
            inputs_embeds = self.token_embedding(input_ids)
        s = inputs_embeds.shape[1]
        x = inputs_embeds + self.position_embedding.weight[:s]                      # modeling_clip.py:36-40
        mask = torch.full((s, s), float("-inf"), device=x.device, dtype=x.dtype).triu(1)   # :44-46 causal
        for l in self.layers:
            x = l(x, mask)
        return self.final_layer_norm(x)                                             # :63


# --------------------------------------------------------------------------------------------
# #15  [3P diffusers 0.14] AutoencoderKL.encode (frozen)  — call site pretrain_e4t.py:598-599
# --------------------------------------------------------------------------------------------
class _VAEAttention(nn.Module):
    """diffusers 0.14 ``AttentionBlock`` (single head at 512 ch): GN -> q,k,v -> softmax -> proj + res."""

    def __init__(self, ch, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.query = nn.Linear(ch, ch)
        self.key = nn.Linear(ch, ch)
        self.value = nn.Linear(ch, ch)
        self.proj_attn = nn.Linear(ch, ch)

    def forward(self, x):
        b, c, hh, ww = x.shape
        h = self.group_norm(x).view(b, c, hh * ww).transpose(1, 2)
        q, k, v = self.query(h), self.key(h), self.value(h)
        att = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
        h = self.proj_attn(att @ v)
        return h.transpose(1, 2).reshape(b, c, hh, ww) + x


class VAEEncoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, groups=32, scaling_factor=0.18215):
        super().__init__()
        boc = tuple(block_out_channels)
        self.scaling_factor = scaling_factor
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down = nn.ModuleList()
        ch = boc[0]
        for i, c in enumerate(boc):
            res = nn.ModuleList([ResnetBlock2D(ch if j == 0 else c, c, 0, groups, 1e-6) for j in range(layers_per_block)])
            ds = Downsample2D(c, padding=0) if i < len(boc) - 1 else None
            self.down.append(nn.ModuleDict(dict(resnets=res)) if ds is None else nn.ModuleDict(dict(resnets=res, downsampler=ds)))
            ch = c
        self.mid_res1 = ResnetBlock2D(ch, ch, 0, groups, 1e-6)
        self.mid_attn = _VAEAttention(ch, groups)
        self.mid_res2 = ResnetBlock2D(ch, ch, 0, groups, 1e-6)
        self.conv_norm_out = nn.GroupNorm(groups, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, 2 * latent_channels, 3, padding=1)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def moments(self, x):
        x = self.conv_in(x)
        for blk in self.down:
            for r in blk["resnets"]:
                x = r(x)
            if "downsampler" in blk:
                x = blk["downsampler"](x)
        x = self.mid_res2(self.mid_attn(self.mid_res1(x)))
        x = self.quant_conv(self.conv_out(F.silu(self.conv_norm_out(x))))
        mean, logvar = x.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def encode_sample(self, x, eps):
        """latent_dist.sample() * scaling_factor with externally supplied noise ``eps``."""
        mean, logvar = self.moments(x)
        return (mean + torch.exp(0.5 * logvar) * eps) * self.scaling_factor


# --------------------------------------------------------------------------------------------
# a11  scheduler + one training step — pretrain_e4t.py:595-654 ; [3P] DDPMScheduler
# --------------------------------------------------------------------------------------------
def ddpm_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2   # "scaled_linear"
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0, noise, t, acp):
    a = acp[t].sqrt().view(-1, 1, 1, 1)
    s = (1 - acp[t]).sqrt().view(-1, 1, 1, 1)
    return a * x0 + s * noise


def get_velocity(x0, noise, t, acp):
    a = acp[t].sqrt().view(-1, 1, 1, 1)
    s = (1 - acp[t]).sqrt().view(-1, 1, 1, 1)
    return a * noise - s * x0


def e4t_losses(unet, e4t_encoder, text_encoder, pixel_values, latents, noise, timesteps, inputs_embeds,
               placeholder_idx: Sequence[int], ctx_for_e4t, class_embed, acp,
               domain_embed_scale=0.1, reg_lambda=0.01, prediction_type="epsilon"):
    """Forward half of the hot loop, pretrain_e4t.py:621-647.  Returns (loss, loss_diff, loss_reg, aux)."""
    bsz = latents.shape[0]
    noisy = add_noise(latents, noise, timesteps, acp)                                           # :621
    enc = unet(noisy, timesteps, ctx_for_e4t.expand(bsz, -1, -1), return_encoder_outputs=True)  # :622-624
    domain = e4t_encoder(pixel_values, enc["down_block_samples"])                               # :626
    domain = class_embed.clone().expand(bsz, -1) + domain_embed_scale * domain                  # :628
    emb = inputs_embeds.clone()
    for i, j in enumerate(placeholder_idx):                                                     # :630-631
        emb[i, j, :] = domain[i]
    ctx = text_encoder(inputs_embeds=emb)                                                       # :634
    pred = unet(noisy, timesteps, ctx)                                                          # :636
    target = noise if prediction_type == "epsilon" else get_velocity(latents, noise, timesteps, acp)  # :638-643
    loss_diff = F.mse_loss(pred.float(), target.float(), reduction="mean")                      # :645
    loss_reg = reg_lambda * domain.pow(2).sum()                                                 # :646
    return loss_diff + loss_reg, loss_diff, loss_reg, dict(pred=pred, domain_embed=domain, enc=enc)


def trainable_parameters(unet, e4t_encoder):
    """pretrain_e4t.py:274-278: encoder params with requires_grad + UNet params whose name contains 'wo'."""
    ps = [p for p in e4t_encoder.parameters() if p.requires_grad]
    ps += [p for n, p in unet.named_parameters() if "wo" in n]
    return ps


def tiny_unet_config(ctx_dim=64):
    """A small 4-level config with the same topology (for tests that must finish in seconds on CPU)."""
    return dict(SD14_UNET_CONFIG, sample_size=16, block_out_channels=(64, 128, 128, 128),
                cross_attention_dim=ctx_dim, attention_head_dim=2)


# --------------------------------------------------------------------------------------------
# N4  inference: [3P diffusers 0.14] AutoencoderKL.decode, DDIMScheduler, and the E4T sampling loop
#     — e4t/pipeline_stable_diffusion_e4t.py:68-250 ; inference.py:75-153 ; pretrain_e4t.py:450-455
# --------------------------------------------------------------------------------------------
class _VAEMid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, 0, groups, 1e-6), ResnetBlock2D(ch, ch, 0, groups, 1e-6)])
        self.attentions = nn.ModuleList([_VAEAttention(ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _VAEUpBlock(nn.Module):
    """UpDecoderBlock2D: layers_per_block + 1 resnets, then nearest x2 + conv (all but the last block)"""

    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, 0, groups, 1e-6) for j in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.upsamplers is not None else x


class _VAEDecoderNet(nn.Module):
    def __init__(self, out_channels, latent_channels, boc, layers_per_block, groups):
        super().__init__()
        rev = tuple(reversed(boc))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _VAEMid(rev[0], groups)
        blocks, ch = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(_VAEUpBlock(ch, c, layers_per_block + 1, groups, i < len(rev) - 1))
            ch = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAEDecoder(nn.Module):
    """AutoencoderKL.decode(latents / scaling_factor).sample — parameter names of the diffusers checkpoint
    (``post_quant_conv``, ``decoder.*``).  Call site: StableDiffusionPipeline.decode_latents, used at
    pipeline_stable_diffusion_e4t.py:226,237."""

    def __init__(self, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, groups=32, scaling_factor=0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _VAEDecoderNet(out_channels, latent_channels, tuple(block_out_channels), layers_per_block, groups)

    def decode(self, latents):
        return self.decoder(self.post_quant_conv(latents / self.scaling_factor))

    def decode_latents(self, latents):
        """-> float32 NHWC in [0, 1] (diffusers decode_latents: (image / 2 + 0.5).clamp(0, 1))"""
        return (self.decode(latents) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()


class DDIMScheduler:
    """[3P diffusers 0.14] DDIMScheduler restated (schedulers/scheduling_ddim.py): the scheduler inference.py and the
    in-training sampler construct (inference.py:64,113 ; pretrain_e4t.py:450).  Defaults = the Stable Diffusion scheduler
    config (scaled_linear 0.00085..0.012, clip_sample False, set_alpha_to_one False, steps_offset 1)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.clip_sample, self.prediction_type = clip_sample, prediction_type
        self.num_inference_steps, self.timesteps = None, None

    def set_timesteps(self, num_inference_steps):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = torch.arange(0, num_inference_steps).mul(ratio).flip(0).long() + self.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, variance_noise=None):
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            model_output = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise NotImplementedError(self.prediction_type)
        if self.clip_sample:
            x0 = x0.clamp(-1, 1)
        variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
        out = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            out = out + std * variance_noise
        return out


@torch.no_grad()
def e4t_sample(unet, e4t_encoder, text_encoder, scheduler, image, inputs_embeds, placeholder_idx, ctx_for_e4t,
               class_embed, latents, num_inference_steps=50, guidance_scale=7.5, domain_embed_scale=0.1, eta=0.0,
               step_noise=None):
    """The denoising loop of StableDiffusionE4TPipeline.__call__ (pipeline_stable_diffusion_e4t.py:160-218) with the
    tokenizer work done by the caller: inputs_embeds (1,77,d) = token embeddings of the prompt, ctx_for_e4t (1,77,d) =
    text_encoder("") output, image (1,3,H,W) in [-1,1], latents (B,4,h,w) ~ N(0,1).  Returns the final latents."""
    scheduler.set_timesteps(num_inference_steps)
    cfg = guidance_scale > 1.0
    latents = latents * scheduler.init_noise_sigma
    bsz = latents.shape[0]
    for i, t in enumerate(scheduler.timesteps):
        x_in = scheduler.scale_model_input(latents, t)
        model_in = torch.cat([x_in] * 2) if cfg else x_in                                          # :181-182
        ctx_e = ctx_for_e4t.expand(bsz, -1, -1)                                                      # :187
        enc = unet(x_in, t.expand(bsz), ctx_e, return_encoder_outputs=True)                         # :189
        domain = e4t_encoder(image.expand(bsz, -1, -1, -1), enc["down_block_samples"])              # :191-192
        domain = class_embed.clone().expand(bsz, -1) + domain_embed_scale * domain                  # :194
        emb = inputs_embeds.expand(bsz, -1, -1).clone()
        emb[:, placeholder_idx, :] = domain                                                          # :195-196
        ctx = text_encoder(inputs_embeds=emb)                                                        # :198
        prompt = torch.cat([ctx_e, ctx]) if cfg else ctx                                             # :199
        pred = unet(model_in, t.expand(model_in.shape[0]), prompt)                                  # :201-206
        if cfg:
            u, c = pred.chunk(2)
            pred = u + guidance_scale * (c - u)                                                      # :209-211
        latents = scheduler.step(pred, t, latents, eta=eta, variance_noise=None if step_noise is None else step_noise[i])  # :214
    return latents
