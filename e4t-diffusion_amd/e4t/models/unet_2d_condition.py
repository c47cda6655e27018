"""UNet2DConditionModel — drop-in for the reference's e4t/models/unet_2d_condition.py (:30-562).

Same constructor keywords (a diffusers UNet config dict), same parameter names (SD checkpoints and
``weight_offsets.pt`` load by key), same ``forward`` signature including the E4T extension
``return_encoder_outputs`` (:423, :517-521) that returns the 13 encoder feature maps.  Internally
everything is NHWC bf16 and every op is a HIP kernel; the only torch work is casting / padding the
4-channel latent input and un-padding the 4-channel output.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple, Union

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from ..utils import AttributeDict
from ..weightoffsets import WOBank
from .cross_attention import CrossAttention, HipAttnProcessor
from .resnet import FMap
from .unet_2d_blocks import UNetMidBlock2DCrossAttn, get_down_block, get_up_block


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


class TimestepEmbedding(nn.Module):
    """[3P diffusers] Linear -> SiLU -> Linear (unet_2d_condition.py:129-135)."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)
        self._p1, self._p2 = Fn.PreparedLinear(self.linear_1.weight), Fn.PreparedLinear(self.linear_2.weight)

    def forward(self, t_emb):
        h = Fn.silu(Fn.linear(t_emb, self.linear_1.weight, self.linear_1.bias, self._p1))
        return Fn.linear(h, self.linear_2.weight, self.linear_2.bias, self._p2)


class _SharedPrefix:
    """Single-use record/replay store for the context-independent prefix of the UNet (see forward()).  A forward whose
    key differs from the recorded one starts a new recording; a forward that replays a recording consumes it."""

    def __init__(self):
        self.key, self.store, self.active, self._replay = None, {}, False, False

    def begin(self, key, enabled):
        self.active = enabled
        if not enabled:
            self.key, self.store = None, {}
            return
        self._replay = key == self.key and len(self.store) > 0
        if not self._replay:
            self.key, self.store = key, {}

    def reuse(self, name, fn):
        if not self.active:
            return fn()
        if name not in self.store:
            self.store[name] = fn()
        return self.store[name]

    def end(self):
        if self._replay:                       # consumed: drop the references (and with them the graph, after backward)
            self.key, self.store, self._replay = None, {}, False


class UNet2DConditionModel(nn.Module):
    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 center_input_sample: bool = False, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
                 up_block_types: Tuple[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 only_cross_attention: Union[bool, Tuple[bool]] = False, block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
                 layers_per_block: int = 2, downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: Optional[int] = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim: Union[int, Tuple[int]] = 8, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, class_embed_type: Optional[str] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, resnet_time_scale_shift: str = "default", time_embedding_type: str = "positional",
                 conv_in_kernel: int = 3, conv_out_kernel: int = 3, **unused):
        super().__init__()
        assert not center_input_sample and flip_sin_to_cos and freq_shift == 0 and act_fn == "silu"
        assert not dual_cross_attention and class_embed_type is None and num_class_embeds is None
        assert resnet_time_scale_shift == "default" and time_embedding_type == "positional" and mid_block_type == "UNetMidBlock2DCrossAttn"
        assert conv_in_kernel == 3 and conv_out_kernel == 3 and downsample_padding == 1 and mid_block_scale_factor == 1
        cfg = dict(locals()); cfg.pop("self"); cfg.pop("unused"); cfg.pop("__class__", None)
        self.config = AttributeDict(cfg)
        self.sample_size, self.in_channels = sample_size, in_channels
        boc = tuple(block_out_channels)
        n = len(boc)
        if isinstance(only_cross_attention, bool):
            only_cross_attention = [only_cross_attention] * n
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * n
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, typ in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            self.down_blocks.append(get_down_block(
                typ, num_layers=layers_per_block, in_channels=in_ch, out_channels=out_ch, temb_channels=temb,
                add_downsample=i != n - 1, resnet_eps=norm_eps, resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                attn_num_head_channels=attention_head_dim[i], use_linear_projection=use_linear_projection,
                only_cross_attention=only_cross_attention[i], upcast_attention=upcast_attention))
        self.mid_block = UNetMidBlock2DCrossAttn(in_channels=boc[-1], temb_channels=temb, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                                                 cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim[-1],
                                                 use_linear_projection=use_linear_projection, upcast_attention=upcast_attention)
        self.up_blocks = nn.ModuleList()
        self.num_upsamplers = 0
        rboc, rheads, roca = list(reversed(boc)), list(reversed(attention_head_dim)), list(reversed(only_cross_attention))
        out_ch = rboc[0]
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, n - 1)]
            last = i == n - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(get_up_block(
                typ, num_layers=layers_per_block + 1, in_channels=in_ch, out_channels=out_ch, prev_output_channel=prev,
                temb_channels=temb, add_upsample=not last, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, attn_num_head_channels=rheads[i], use_linear_projection=use_linear_projection,
                only_cross_attention=roca[i], upcast_attention=upcast_attention))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self._pin, self._pout = Fn.PreparedConv(self.conv_in.weight), Fn.PreparedConv(self.conv_out.weight)
        self._groups, self._eps, self._temb_in = norm_num_groups, norm_eps, boc[0]
        # Weight-offset banks in gradient-finalisation order: the up-block heads are final
        # first in the backward, the mid/down heads (shared by both UNet passes) last.
        self._prefix = _SharedPrefix()
        from .resnet import ResnetBlock2D
        self._resblocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        self._temb_cat, self._temb_rb = None, None
        self.share_prefix = False          # enabled by `with unet.shared_prefix():` around the two passes of one step
        self.on_up_backward_done = None    # trainer hook: called in the backward once every gradient of up_blocks / conv_norm_out / conv_out is final
        self.wo_banks = [WOBank("up"), WOBank("mid_down")]
        for mod in self.up_blocks.modules():
            if isinstance(mod, CrossAttention):
                mod.register_bank(self.wo_banks[0])
        for part in (self.mid_block, self.down_blocks):
            for mod in part.modules():
                if isinstance(mod, CrossAttention):
                    mod.register_bank(self.wo_banks[1])

    # ---------------------------------------------------------------------- reference-surface helpers
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def attn_processors(self) -> Dict[str, object]:
        return {f"{n}.processor": m.processor for n, m in self.named_modules() if isinstance(m, CrossAttention)}

    def set_attn_processor(self, processor):
        for n, m in self.named_modules():
            if isinstance(m, CrossAttention):
                m.set_processor(processor[f"{n}.processor"] if isinstance(processor, dict) else processor)

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        self.set_attn_processor(HipAttnProcessor())

    def set_use_memory_efficient_attention_xformers(self, valid: bool, attention_op=None):
        self.set_attn_processor(HipAttnProcessor())

    def _clear_rb(self):
        for r in self._resblocks:
            r._rb = None

    def _time_emb_proj_all(self, temb_act):
        """One GEMM for the time-embedding projections of all ResBlocks (functional.TimeEmbProjAllFn); None when any of
        them is trainable (tuning) — the blocks then run their own Linear."""
        lins = [r.time_emb_proj for r in self._resblocks]
        if any(l is None or l.weight.requires_grad or l.bias is None or l.bias.requires_grad for l in lins):
            return None
        key = (tuple(l.weight.data_ptr() for l in lins), temb_act.dtype, temb_act.device)
        if self._temb_cat is None or self._temb_cat[0] != key:
            w = torch.cat([l.weight.detach() for l in lins], dim=0).to(temb_act.dtype).contiguous()      # [sum Cout, temb]
            b = torch.cat([l.bias.detach().float() for l in lins], dim=0).contiguous()
            splits, o = [], 0
            for l in lins:
                splits.append((o, l.out_features))
                o += l.out_features
            self._temb_cat = (key, w, w.t().contiguous(), b, tuple(splits))
        _, w, wt, b, splits = self._temb_cat
        return Fn.TimeEmbProjAllFn.apply(temb_act, w, wt, b, splits)

    def shared_prefix(self):
        """Context manager: forwards issued inside it on the SAME (sample, timestep) tensors share the context-independent
        prefix (see forward()).  Use it around the encoder pass + full pass of one training step, before the backward."""
        unet = self

        class _Ctx:
            def __enter__(self_inner):
                unet.share_prefix = True
                unet._prefix.begin(None, False)
                return unet

            def __exit__(self_inner, *exc):
                unet.share_prefix = False
                unet._prefix.begin(None, False)      # drops every recorded tensor
                return False
        return _Ctx()

    # ---------------------------------------------------------------------- forward
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True, return_encoder_outputs=False):
        assert attention_mask is None and class_labels is None and timestep_cond is None and not cross_attention_kwargs
        assert down_block_additional_residuals is None and mid_block_additional_residual is None
        act = ops.ACT
        B, Cin, H, W = sample.shape
        assert H % (2 ** self.num_upsamplers) == 0 and W % (2 ** self.num_upsamplers) == 0
        dev = sample.device
        be = ops.backend()
        # the mid/down bank (shared by both passes of a step) opens now; the up bank only when the up blocks are reached, so
        # that its backward node is YOUNGER than every mid/down node of this pass: autograd runs ready nodes youngest first,
        # i.e. the up bank's gradients are final — and its all-reduce can start — the moment the backward leaves the up blocks
        self.wo_banks[1].begin(dev)
        # 1. time embedding (unet_2d_condition.py:441-468)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.int64, device=dev)
        elif t.dim() == 0:
            t = t[None].to(dev)
        t = t.expand(B)
        # Shared prefix (restructuring 3): the encoder pass and the full pass of one training step get the
        # same noisy latents and timesteps (pretrain_e4t.py:624,636), so everything that does not see the text context —
        # time embedding, conv_in, down_blocks.0.resnets.0, and attentions.0 up to and including its self-attention — is
        # bit-identical in both.  The first pass records those tensors (with their autograd graph), the second one reuses
        # them: gradients of both passes add at the shared nodes exactly as they add at the shared parameters.
        prefix = self._prefix
        key = (sample.data_ptr(), sample._version, tuple(sample.shape), timestep.data_ptr() if torch.is_tensor(timestep) else timestep,
               timestep._version if torch.is_tensor(timestep) else 0, torch.is_grad_enabled(), ops.weights_epoch(), self.training)
        prefix.begin(key, self.share_prefix and torch.is_tensor(timestep))
        prefix.reuse("inputs", lambda: (sample, timestep))     # keeps them alive: the key's addresses cannot be recycled

        def stem():
            emb = self.time_embedding(be.timestep_embedding(t, self._temb_in))
            temb_act = Fn.silu(emb)      # every ResBlock consumes silu(emb): evaluate it once
            self._temb_rb = self._time_emb_proj_all(temb_act)
            # 2. conv_in: 4 latent channels zero-padded to one 64-wide K tile
            x = torch.zeros((B * H * W, 64), dtype=act, device=dev)
            x[:, :Cin] = sample.permute(0, 2, 3, 1).reshape(B * H * W, Cin)
            return temb_act, Fn.conv3x3(x, self.conv_in.weight, self.conv_in.bias, self._pin, (B, H, W, H, W))
        self._temb_rb = None
        temb_act, x0, rbs = prefix.reuse("stem", lambda: stem() + (self._temb_rb,))
        for r, rb in zip(self._resblocks, rbs or [None] * len(self._resblocks)):
            r._rb = rb
        ctx = encoder_hidden_states.to(act).contiguous()
        m = FMap(x0, B, H, W)
        # 3. down
        skips = (m,)
        for i, blk in enumerate(self.down_blocks):
            if i == 0 and prefix.active and blk.has_cross_attention:
                m, outs = blk.forward_nhwc(m, temb_act, ctx, prefix)
            else:
                m, outs = blk.forward_nhwc(m, temb_act, ctx)
            skips += outs
        prefix.end()
        # 4. mid
        m = self.mid_block.forward_nhwc(m, temb_act, ctx)
        if return_encoder_outputs:
            self._clear_rb()
            return dict(down_block_samples=tuple(s.nchw() for s in skips + (m,)))
        # 5. up
        self.wo_banks[0].begin(dev)
        cb = self.on_up_backward_done
        if cb is not None and torch.is_grad_enabled() and m.x.requires_grad:
            m.x.register_hook(lambda g: (cb(), None)[1])      # gradient w.r.t. the mid-block output: every up-block gradient is final
        for blk in self.up_blocks:
            k = len(blk.resnets)
            s, skips = skips[-k:], skips[:-k]
            m = blk.forward_nhwc(m, s, temb_act, ctx)
        # 6. out: GN + SiLU + conv_out straight to fp32
        h = Fn.group_norm(m.x, None, self.conv_norm_out.weight, self.conv_norm_out.bias, B, H * W, self._groups, self._eps, True)
        y = Fn.conv3x3(h, self.conv_out.weight, self.conv_out.bias, self._pout, (B, H, W, H, W), out_f32=True)
        out = y.view(B, H, W, -1).permute(0, 3, 1, 2)
        self._clear_rb()
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)
