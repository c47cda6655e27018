"""One E4T pre-training step (reference: pretrain_e4t.py:595-654) on the native modules.

``E4TTrainer`` owns what the reference spreads over accelerate + torch.optim:
  * flat fp32 parameter / gradient / Adam-moment buffers for exactly the tensors the reference optimises
    (pretrain_e4t.py:274-278: encoder parameters with requires_grad + UNet parameters whose name contains
    "wo"); parameters are re-pointed to views of the flat buffer, their .grad to views of the flat gradient,
    so the kernels that produce gradients write their final location directly;
  * ONE fused AdamW launch over the flat buffers (torch.optim.AdamW defaults, :387-392);
  * data-parallel gradient averaging: one process per GPU, RCCL all-reduce of the flat gradient in
    finalisation-ordered buckets on a side stream (the reference gets this implicitly from DDP, :410-412,648).
The step itself is the reference's: VAE encode -> noise -> UNet encoder pass -> E4T encoder -> embed inject ->
text encoder -> UNet full pass -> MSE + lambda * |e|^2 -> backward -> AdamW.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import contextlib
import os
import time
import warnings

import torch
import torch.nn.functional as F

from . import functional as Fn
from . import ops

f32 = torch.float32


def ddpm_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012, device=None):
    """[3P diffusers] DDPMScheduler(scaled_linear) as used at pretrain_e4t.py:235."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=f32, device=device) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def select_trainable(unet, encoder, tuning=False, text_encoder=None):
    """pre-training (pretrain_e4t.py:274-278): encoder params with requires_grad + UNet params whose name contains "wo";
    every other UNet parameter is frozen (the reference never reads their grads).
    domain tuning (tuning_e4t.py:139-147): the whole UNet + the encoder's trainable parameters (+ the text encoder's when
    --train_text_encoder left them trainable).  Order = flat-buffer order: [E4T encoder | text encoder | UNet]."""
    for n, p in unet.named_parameters():
        p.requires_grad_(tuning or "wo" in n)
    named = [(f"e4t_encoder.{n}", p) for n, p in encoder.named_parameters() if p.requires_grad]
    if text_encoder is not None:
        named += [(f"text_encoder.{n}", p) for n, p in text_encoder.named_parameters() if p.requires_grad]
    named += [(f"unet.{n}", p) for n, p in unet.named_parameters() if tuning or "wo" in n]
    return named


class FlatParams:
    """Flat fp32 storage for a list of parameters (64-float aligned segments) + matching gradient buffer."""

    def __init__(self, params: Sequence[torch.nn.Parameter], device):
        self.params = list(params)
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.offsets, self.numel = offs, off
        self.data = torch.zeros(off, dtype=f32, device=device)
        self.grad = torch.zeros(off, dtype=f32, device=device)
        for p, o in zip(self.params, offs):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def view(self, p_index, buf):
        p, o = self.params[p_index], self.offsets[p_index]
        return buf[o:o + p.numel()].view(p.shape)


class _AlwaysFalse(dict):
    """_exchange_ok after a ragged batch was seen: every batch size answers False without another collective"""

    def get(self, key, default=None):
        return False


def _runs_beside(main, side, device) -> bool:
    """does a kernel launched on `side` execute while `main` is busy?  (False: the two streams share a hardware queue)"""
    x = torch.zeros(64, device=device)
    e_main, e_side = torch.cuda.Event(), torch.cuda.Event()
    torch.cuda.synchronize(device)
    with torch.cuda.stream(main):
        torch.cuda._sleep(20_000_000)          # a spin kernel, 9.6 ms on MI355X; the probe stops polling as soon as it knows
        e_main.record(main)
    with torch.cuda.stream(side):
        x.add_(1)
        e_side.record(side)
    ok = False
    while not e_main.query():
        if e_side.query():
            ok = True
            break
    torch.cuda.synchronize(device)
    return ok


class E4TTrainer:
    def __init__(self, unet, e4t_encoder, text_encoder, vae, *, lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 domain_embed_scale=0.1, reg_lambda=0.01, prediction_type="epsilon", class_token_id=0,
                 empty_prompt_ids: Optional[torch.Tensor] = None, process_group=None, device=None, tuning=False,
                 max_grad_norm: Optional[float] = None, head_factor_exchange=True, collectives="torch"):
        self.unet, self.encoder, self.text_encoder, self.vae = unet, e4t_encoder, text_encoder, vae
        # data parallel: all-gather the two small factors of the 129-slot head's weight gradient instead of all-reducing the 845 MB
        # stack (_exchange_head_factors); needs the same per-rank batch size on every rank.  False = the stack rides the all-reduce.
        self.head_factor_exchange = bool(head_factor_exchange)
        self.device = device or next(unet.parameters()).device
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.scale, self.reg_lambda, self.pred_type = domain_embed_scale, reg_lambda, prediction_type
        self.share_prefix = True     # compute the context-independent UNet prefix once for the step's two passes
        self.overlap_vision = True           # the frozen CLIP-ViT on a side stream (under the encoder pass / one batch ahead)
        self._side, self._vision, self._vision_event = None, None, None
        # Next-batch prefetch of the step's FROZEN, weight-independent front ends (prefetch()): "vit+vae" (default) = CLIP-ViT tokens and VAE
        # latents of batch i+1 are computed on the side stream under step i's backward; "vit" / "vae" = one of them (the ViT then runs
        # in its own step, on the side stream under the UNet encoder pass); "0" = off (round 3 behaviour).  Measured (round 4, B = 16,
        # one box, +-0.1 ms run to run): off 104.6, vae 102.8, vit+vae 102.2 ms per step.  Starting the side work with the step instead of
        # with its backward measured 101.1-101.5 against 100.3 ms (another box) and was removed; so was evaluating the next step's W_eff
        # at the tail of the step (102.13 / 102.23 against 102.12 / 102.35 ms).
        self.prefetch_mode = os.environ.get("E4T_PREFETCH", "vit+vae")
        self._next_px, self._pref = None, {}          # announced batch; finished / running prefetches by id(pixel tensor)
        self._next_eps = None                         # the VAE's sampling noise for the announced batch (tests); None = drawn when the prefetch starts
        # whole-step HIP graph (enable_step_graph): signature -> captured graph + its static tensors; device copy of AdamW's
        # step-dependent scalars
        self._step_graph_on = False
        self._step_graphs, self._seen_sigs = {}, set()
        self._hyper, self._hyper_ring, self._hyper_i = None, [], 0
        self._capturing = self._graph_failed = False
        self._train_stream, self._train_stream_probed = None, False      # _training_stream()
        self.deferred_region = None          # the region whose all-reduce the last synchronising step ran AdamW of the others under ("D" / None)
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        # E4T_FORCE_COMM=1: run the collective path even in a 1-rank group (exercises the RCCL calls / stream ordering on one GPU)
        self._comm = self.world > 1 or (os.environ.get("E4T_FORCE_COMM") == "1" and torch.distributed.is_available() and torch.distributed.is_initialized())
        # "torch": the gradient collectives are torch.distributed's on `process_group` (the launcher's communicator).  "library": they go
        # through the C ABI's own RCCL communicator (e4t/comm.py, include/e4t_hip.h e4t_comm_*) on the library's stream — what a host
        # without torch.distributed calls; the handful of control-plane collectives (batch-size / finiteness flags) stay on the group.
        if collectives not in ("torch", "library"):
            raise ValueError(f"collectives={collectives!r}: 'torch' or 'library'")
        self._lib_comm = None
        if collectives == "library" and self._comm:
            from .comm import LibraryComm
            self._lib_comm = LibraryComm.from_process_group(process_group)
        self.step_count = 0
        self.acp = ddpm_alphas_cumprod(device=self.device)
        self.max_grad_norm = max_grad_norm
        self.tuning = tuning
        self._armed = False
        named = select_trainable(unet, e4t_encoder, tuning=tuning, text_encoder=text_encoder)
        self.text_trainable = any(n.startswith("text_encoder.") for n, _ in named)
        # order: the stacked first_linears weights, then their biases (contiguous stacks), then the rest
        fl_w = [p for n, p in named if ".first_linears." in n and n.endswith(".weight")]
        fl_b = [p for n, p in named if ".first_linears." in n and n.endswith(".bias")]
        rest = [p for n, p in named if ".first_linears." not in n]
        # make the per-slot segments exactly contiguous: numel of each is a multiple of 64 for hid % 8 == 0
        self.flat = FlatParams(fl_w + fl_b + rest, self.device)
        n = len(fl_w)
        if n:
            hid = fl_w[0].shape[0]
            assert (hid * hid) % 64 == 0 and hid % 64 == 0, "first_linears stacks need 64-float aligned slots"
            o_w, o_b = self.flat.offsets[0], self.flat.offsets[n]
            st = lambda buf, o, shape: buf[o:o + math.prod(shape)].view(shape)
            e4t_encoder.adopt_stacks(st(self.flat.data, o_w, (n, hid, hid)), st(self.flat.data, o_b, (n, hid)),
                                     st(self.flat.grad, o_w, (n, hid, hid)), st(self.flat.grad, o_b, (n, hid)))
            self._stack_grad_off = o_w
            self._stack_shape = (n, hid, hid)
            e4t_encoder._stack_grad_is_zero = True          # the flat gradient starts as zeros (zero_grad keeps the mark up to date)
        if not n:
            self._stack_grad_off = -1
            self._stack_shape = None
        # The head's stacked weight gradient (845 of the 1500 MB at SD sizes) is a rank-B product: a synchronising step hands its two
        # factors to AdamW (e4t_adamw_rank) instead of writing, reading and clearing the stack.  Off with a gradient clip (the norm is
        # taken over the materialised gradient) and while micro-batches accumulate.
        self.factored_head_update = True
        self._exchange_ok = {}               # per-rank batch size -> every rank has it (checked with one collective at first use)
        self._head_factors = None
        self._accum_pending = False          # a non-synchronising step has left gradients behind (graph replay and the factored update assume none)
        self._setup_overlap(named, n)
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        # replica consistency (the reference gets it from the DDP constructor, which broadcasts rank 0's parameters and buffers when
        # accelerate wraps the models, pretrain_e4t.py:410-412): every rank starts from rank 0's weights — trainable AND frozen —
        # whatever its own seed / checkpoint read produced, and a checksum of the trainable state is compared every
        # `replica_check_every` optimiser steps (E4T_REPLICA_CHECK_EVERY, default 100, 0 = off): diverged replicas abort loudly
        # instead of training on silently.
        self.replica_check_every = int(os.environ.get("E4T_REPLICA_CHECK_EVERY", "100"))
        self.comm_timing = None          # bench.py sets a dict: per-region enqueue times and the exposed wait of the last step
        if self.world > 1:
            self.sync_replicas(include_frozen=True)
        ops.bump_weights_epoch()
        # constants of the loop (pretrain_e4t.py:561-583); with a trainable text encoder they are re-evaluated every step, detached,
        # as tuning_e4t.py:276-284 does
        self.class_token_id = class_token_id
        self.empty_prompt_ids = (empty_prompt_ids if empty_prompt_ids is not None else torch.zeros((1, 77), dtype=torch.long)).to(self.device)
        self._refresh_text_constants()

    # ---- replica consistency -----------------------------------------------------------------------------------------------
    def sync_replicas(self, include_frozen=False, include_moments=False):
        """rank 0's state -> every rank: the flat trainable buffer (+ Adam moments on resume), optionally every frozen parameter
        and buffer of the models (DDP-constructor semantics).  No-op without a process group."""
        if self.world <= 1:
            return
        bc = lambda t: torch.distributed.broadcast(t, src=torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        bc(self.flat.data)
        if include_moments:
            bc(self.exp_avg); bc(self.exp_avg_sq)
            cnt = torch.tensor([self.step_count], dtype=torch.int64, device=self.device)
            bc(cnt)
            self.step_count = int(cnt)
        if include_frozen:
            mine = {id(p) for p in self.flat.params}
            for m in (self.unet, self.encoder, self.text_encoder, self.vae):
                if m is None:
                    continue
                with torch.no_grad():
                    for t in list(m.parameters()) + list(m.buffers()):
                        if id(t) not in mine and t.numel() and (t.is_floating_point() or t.dtype in (torch.int32, torch.int64)):
                            # broadcast INTO the tensor itself (not .data): the in-place write bumps its _version, which is what the
                            # bf16 compute copies of frozen weights (PreparedLinear / PreparedConv / the text encoder's fused
                            # qkv and captured graphs) are keyed on — a forward that ran before this trainer was built must not
                            # leave ranks > 0 computing with their pre-broadcast copies
                            bc(t)
        ops.bump_weights_epoch()

    def check_replicas(self):
        """every rank must hold bit-identical trainable state (deterministic kernels + the same all-reduced gradient): compare a
        checksum vector — sum and sum of squares of the parameters, of exp_avg and of exp_avg_sq — across ranks (MIN == MAX);
        raises on every rank when they differ.  A non-finite state is reported as such, not as divergence (NaN != NaN)."""
        if self.world <= 1:
            return True
        bufs = (self.flat.data, self.exp_avg, self.exp_avg_sq)
        if self.flat.data.is_cuda:
            be = ops.backend()
            c = torch.stack([x for b in bufs for x in (b.sum(dtype=torch.float64), be.sumsq(b).double())])
        else:
            c = torch.stack([x for b in bufs for x in (b.double().sum(), b.double().pow(2).sum())])
        finite = torch.isfinite(c).all().to(torch.int32)
        torch.distributed.all_reduce(finite, op=torch.distributed.ReduceOp.MIN, group=self.pg)
        if not bool(finite):
            raise RuntimeError(f"non-finite training state at optimiser step {self.step_count} (a NaN / Inf in the parameters or Adam moments "
                               f"of at least one rank): checksums {c.tolist()}")
        lo, hi = c.clone(), c.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=self.pg)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            raise RuntimeError(f"data-parallel replicas diverged at optimiser step {self.step_count}: checksum range {lo.tolist()} .. {hi.tolist()} "
                               "(ranks no longer hold identical parameters — different seeds / checkpoints without the start-up broadcast, "
                               "or a non-deterministic gradient path)")
        return True

    def prepare(self, batch_size):
        """One-off work that should not land inside the first training step: the frozen text encoder's HIP graphs for this batch size
        are captured here, before the caller starts its data-loader thread (a capture and another thread's allocations do not mix)."""
        te = self.text_encoder
        if hasattr(te, "prepare_graphs") and not self.text_trainable:
            te.prepare_graphs(batch_size, self.device)

    def _refresh_text_constants(self):
        with torch.no_grad():
            emb = self.text_encoder.get_input_embeddings()
            self.class_embed = emb(torch.tensor([self.class_token_id], device=self.device))[0].float()
            self.ctx_for_e4t = self.text_encoder(input_ids=self.empty_prompt_ids)[0].detach()

    # ------------------------------------------------------------------------------------------------
    def add_noise(self, x0, noise, t):
        a = self.acp[t].sqrt().view(-1, 1, 1, 1)
        s = (1 - self.acp[t]).sqrt().view(-1, 1, 1, 1)
        return a * x0 + s * noise

    def losses(self, pixel_values, latents, noise, timesteps, input_ids, placeholder_idx):
        """Forward half of the step (pretrain_e4t.py:616-647).  Returns (loss, loss_diff, loss_reg)."""
        B = latents.shape[0]
        te = self.text_encoder
        if self.text_trainable:
            self._refresh_text_constants()
        with torch.set_grad_enabled(self.text_trainable and torch.is_grad_enabled()):     # the embedding table trains with the text encoder
            inputs_embeds = te.get_input_embeddings()(input_ids)
        noisy = self.add_noise(latents, noise, timesteps)
        # both UNet passes see the same (noisy, timesteps): the context-independent prefix is computed once
        share = self.unet.shared_prefix() if (self.share_prefix and hasattr(self.unet, "shared_prefix")) else contextlib.nullcontext()
        # The frozen CLIP-ViT only needs the image: run it on a side stream under the UNet encoder pass, whose low-resolution
        # levels leave CUs idle (one process per GPU, two HIP streams; joined before the E4T head needs the tokens).  Measured:
        # -2.4 ms/step here; launching it even earlier, under the VAE encode (chip already full), gains nothing.
        joined = self._vision is not None           # prefetched by the previous step: joined below, where the tokens are consumed
        vision, self._vision = (self._vision if joined else self._launch_vision(pixel_values)), None
        vis_ev, self._vision_event = self._vision_event, None
        with share:
            enc = self.unet(noisy, timesteps, self.ctx_for_e4t.expand(B, -1, -1), return_encoder_outputs=True)
            if vision is not None and not joined:
                torch.cuda.current_stream().wait_stream(self._side)
                for t in vision:
                    t.record_stream(torch.cuda.current_stream())
            elif vision is not None and vis_ev is not None:      # prefetched tokens: the ViT's tail ran under the encoder pass above
                torch.cuda.current_stream().wait_event(vis_ev)
                for t in vision:
                    t.record_stream(torch.cuda.current_stream())
            if vision is not None:
                domain = self.encoder(x=pixel_values, unet_down_block_samples=enc["down_block_samples"], vision=vision)
            else:
                domain = self.encoder(x=pixel_values, unet_down_block_samples=enc["down_block_samples"])
            domain = self.class_embed[None, :].expand(B, -1) + self.scale * domain
            emb = inputs_embeds.clone()
            emb[torch.arange(B, device=emb.device), placeholder_idx] = domain.to(emb.dtype)
            ctx = te(inputs_embeds=emb)[0]
            pred = self.unet(noisy, timesteps, ctx).sample
        if self.pred_type == "epsilon":
            target = noise
        else:
            a = self.acp[timesteps].sqrt().view(-1, 1, 1, 1)
            s = (1 - self.acp[timesteps]).sqrt().view(-1, 1, 1, 1)
            target = a * noise - s * latents
        loss_diff = F.mse_loss(pred.float(), target.float(), reduction="mean")
        loss_reg = self.reg_lambda * domain.pow(2).sum()
        return loss_diff + loss_reg, loss_diff, loss_reg

    def _vision_applicable(self, pixel_values):
        return bool(self.overlap_vision and pixel_values.is_cuda and hasattr(self.encoder, "encode_vision") and self.encoder.vision_is_frozen())

    def _launch_vision(self, pixel_values):
        """Start the frozen CLIP-ViT on the side stream (None when not applicable: CPU, trainable ViT, switched off)."""
        if not self._vision_applicable(pixel_values):
            return None
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = self._new_side_stream(pixel_values.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            return self.encoder.encode_vision(pixel_values)

    def _new_side_stream(self, device):
        """(HIP offers two stream priorities here, -1 and 0; running the step at -1 or the side stream "low" measured no difference:
        101.3 vs 101.1-101.5 ms, profiles/r04_ab; the side stream at HIGH priority: 103.05 / 103.51 vs 102.53 / 102.56 ms, round 5.  A side stream confined to 64 / 128 of the 256 CUs with hipExtStreamCreateWithCUMask —
        round-4 review item 7 — measured 177.4 / 137.9 ms per step against 104.8: the side work is ~35 ms of full-chip time and a slice of
        the chip stretches it past the backward it hides under; profiles/r05_ab/r05a_side_cus*.json.)

        A new HIP stream is NOT guaranteed to run beside the current one: the runtime maps streams onto a handful of hardware queues, and
        two streams that land on the same queue serialise.  Which stream objects share the current stream's queue depends on how many
        streams the process created before — with an initialised RCCL process group the stream this function used to return shared it:
        no overlap at all, 105.0 against 100.3 ms per step on one GPU (profiles/r05_ab/comm_path_one_gpu.txt).  So candidates are
        PROBED: a spin kernel on the current stream, a one-element kernel on the candidate, and the candidate is taken when its kernel
        finishes while the spin is still running (tools/probe/stream_queues.py prints the pattern)."""
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream(device)
        rejected = []          # kept alive until the choice is made, so the next candidate is a different stream
        for _ in range(8):
            s = torch.cuda.Stream(device=device)
            if _runs_beside(main, s, device):
                return s
            rejected.append(s)
        warnings.warn("no HIP stream that runs beside the training stream was found: the next-batch prefetch will not overlap with the step")
        return rejected[0]

    # ---- next-batch prefetch of the frozen front ends ---------------------------------------------------------------------
    # The CLIP-ViT tower (and the VAE encoder) of a step depend on the step's IMAGES only — not on any weight the optimiser
    # touches — so, like the loader's H2D copy, they can run one batch ahead: a caller that already holds the next batch
    # (DeviceLoader prefetches two deep; bench.py's pool) announces it with prefetch(), and the current step starts those
    # encoders for it on the side stream when its backward begins.  The backward is where the chip has room: the 8 x 8 / 16 x 16
    # levels leave CUs idle and the GroupNorm / LayerNorm / GEGLU / AdamW passes leave the matrix cores idle, while under the
    # forward the side stream mostly displaced main-stream work (round 3: -2.4 of the ViT's ~10 ms).  Every step still computes one
    # ViT (+ VAE) pass — for the batch after it — so the work per step is unchanged; nothing is cached across steps.
    def prefetch(self, pixel_values_next, vae_eps=None):
        """Announce the images of the NEXT train_step (the same tensor object must then be passed to it).  No-op when the ViT is
        trainable, on CPU, or with E4T_PREFETCH=0."""
        self._next_px = pixel_values_next if (self.prefetch_mode != "0" and pixel_values_next is not None and pixel_values_next.is_cuda) else None
        self._next_eps = vae_eps              # the VAE's sampling noise for that batch (tests); None = drawn when the prefetch starts

    def _start_prefetch(self):
        """Start the frozen front ends of the announced batch on the side stream: the VAE encoder FIRST, then the CLIP-ViT, each with its own
        event.  The next step needs the latents at once (noise -> UNet encoder pass) but the ViT tokens only when the E4T head runs, after
        the whole encoder pass: with one event behind both (round 4) the main stream sat idle at every step boundary until the LAST side
        kernel had finished (tools/idle_report.py, profiles/r04_idle_report.txt: 16.9 of 112 ms per step with only the side stream
        running); now it waits for the latents only and the tail of the ViT runs under the next step's encoder pass."""
        px, self._next_px = self._next_px, None
        if px is None:
            return
        pref = dict(px=px, vision=None, latents=None, done=None, done_vision=None)
        want_vae = "vae" in self.prefetch_mode and self.vae is not None
        want_vit = "vit" in self.prefetch_mode and self._vision_applicable(px)
        if not (want_vae or want_vit):
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = self._new_side_stream(px.device)
        if want_vae:
            hl, wl = px.shape[2] // 8, px.shape[3] // 8
            eps, self._next_eps = self._next_eps, None
            vae_eps = eps if eps is not None else torch.randn((px.shape[0], 4, hl, wl), device=px.device)      # drawn on the main stream, in step order
        self._side.wait_stream(main)            # the main stream's position: the start of this step's backward
        if want_vae:
            with torch.cuda.stream(self._side), torch.no_grad():
                pref["latents"] = self.encode_latents(px, vae_eps)
            pref["done"] = self._side.record_event()
        if want_vit:
            with torch.cuda.stream(self._side):
                pref["vision"] = self.encoder.encode_vision(px)
            pref["done_vision"] = self._side.record_event()
        px.record_stream(self._side)        # an announced batch that is dropped untrained must not be freed under the side stream
        if not getattr(self, "_prefetch_warm", False):
            # the first pass of a frozen model also writes its one-time bf16 weight copies (PreparedConv / VAEEncoder._prepare): a
            # main-stream consumer of those copies must not overtake them, so the main stream joins the side stream this once
            main.wait_stream(self._side)
            self._prefetch_warm = True
        if len(self._pref) > 4:                 # announced batches that were never trained on
            self._pref.pop(next(iter(self._pref)))
        self._pref[id(px)] = pref

    def _take_prefetched(self, pixel_values):
        """(vision, latents, vision event) computed for exactly this tensor by the previous step, else (None, None, None).  Joins the side
        stream at the LATENTS' event only; the caller waits for `vision event` where the tokens are consumed (losses)."""
        pref = self._pref.pop(id(pixel_values), None)
        if pref is None or pref["px"] is not pixel_values:
            return None, None, None
        main = torch.cuda.current_stream()
        if pref["latents"] is not None:
            main.wait_event(pref["done"])      # (wait_stream would also wait for the ViT behind it, and for a prefetch of the batch after this one)
            pref["latents"].record_stream(main)
        return pref["vision"], pref["latents"], pref["done_vision"]

    def encode_latents(self, pixel_values, vae_eps):
        w = next(self.vae.parameters())
        return self.vae.encode_sample(pixel_values.to(w.dtype), vae_eps).float()

    # ---- data-parallel gradient averaging, overlapped with the backward ---------------------------------------------
    # The flat gradient is laid out [E4T encoder | UNet down / mid (+ conv_in, time embedding) | UNet up (+ conv_norm_out,
    # conv_out)] — in pre-training the UNet part holds only the weight offsets, in tuning every UNet parameter.  The order in
    # which the regions become final in the backward:
    #   U  when the full-pass backward leaves the up blocks: the UNet calls `on_up_backward_done` from a hook on the gradient
    #      of the mid-block output, by which time the up bank's own backward node has run (it is created right before the up
    #      blocks, so autograd — youngest ready node first — runs it before any mid-block node);
    #   H  when the encoder's backward is complete (hook on the gradient of its pooled-UNet-feature input; a trainable ViT is
    #      younger than that node and therefore done): before the whole encoder-pass UNet backward;
    #   W  the stacked weights of the head's 129 linears, the front of the encoder's part (845 of its 920 MB at SD sizes): final with H,
    #      and normally NOT reduced at all — the head's backward gathers every rank's factors and writes the global sum itself
    #      (_exchange_head_factors); reduced with H when that did not happen (accumulated micro-batches, exchange switched off);
    #   D  at the very end (the mid/down parameters are shared by both UNet passes).
    # Each region's all-reduce is enqueued on RCCL's stream the moment it is final (async_op), in 256 MB buckets; the step only
    # waits for the handles before the gradient clip / AdamW.
    def _setup_overlap(self, named, n_first):
        self._works, self._done = [], set()
        self.regions = None
        up_bank, md_bank = self.unet.wo_banks
        up_bank.on_backward_done = md_bank.on_backward_done = None          # a previous trainer's announcements, if any
        self.unet.on_up_backward_done = None
        self.encoder.on_backward_done = None
        self.encoder.exchange_head_factors = None
        self.encoder.take_head_factors = self._take_head_factors if n_first else None
        if not self._comm:
            return
        params = self.flat.params
        pid = {id(p): n for n, p in named}          # `named` order == flat order only after the first_linears re-ordering
        flat_names = [pid[id(p)] for p in params]
        first_unet = next(i for i, n in enumerate(flat_names) if n.startswith("unet."))
        first_text = next((i for i, n in enumerate(flat_names) if n.startswith("text_encoder.")), first_unet)
        first_up = next(i for i, n in enumerate(flat_names) if n.startswith("unet.up_blocks."))
        tail = ("unet.up_blocks.", "unet.conv_norm_out.", "unet.conv_out.")
        assert all(n.startswith(tail) for n in flat_names[first_up:]) and all(not n.startswith("unet.") for n in flat_names[:first_unet])
        assert not any(n.startswith(tail) for n in flat_names[first_unet:first_up])
        assert all(n.startswith("e4t_encoder.") for n in flat_names[:first_text]) and all(n.startswith("text_encoder.") for n in flat_names[first_text:first_unet])
        o = self.flat.offsets
        # T (trainable text encoder, tuning_e4t.py --train_text_encoder): its token embedding is the OLDEST autograd node of the step
        # (inputs_embeds = embedding(input_ids) is evaluated first), so its gradient is final only when the backward ends — T has
        # no hook and is reduced by the sweep after the backward, never together with H
        w_end = o[n_first] if n_first else 0          # the first_linears weight stack leads the flat buffer (constructor)
        self.regions = dict(W=(0, w_end), H=(w_end, o[first_text]), T=(o[first_text], o[first_unet]), D=(o[first_unet], o[first_up]),
                            U=(o[first_up], self.flat.numel))
        self._up_events = 0

        def up_event(*_):
            # two announcements per step: the up bank's backward node and the mid-output gradient hook
            self._up_events += 1
            if self._up_events == 2:
                self._reduce_region("U")
        up_bank.on_backward_done = up_event
        self.unet.on_up_backward_done = up_event
        md_bank.on_backward_done = lambda b: self._reduce_region("D")
        self.encoder.on_backward_done = lambda: (self._reduce_region("W"), self._reduce_region("H"))
        if n_first and self.head_factor_exchange:
            self.encoder.exchange_head_factors = self._exchange_head_factors

    def _exchange_head_factors(self, gb, Z):
        """dW_i = gb^T z_i summed over the ranks = one product over all ranks' rows: gather [gb | Z] (B x (1 + 129) x 1280 bf16, 5.3 MB per
        rank at B = 16) instead of all-reducing the 845 MB fp32 stack — 56 % of the step's all-reduce bytes, and every rank forms the
        sum from the same gathered rows in the same order (bitwise-equal replicas).  Called by the head's backward on the first write
        of the stack in a synchronising step; returns every rank's rows (rank order) and marks region W as already global, or None."""
        if not self._armed or self.regions is None or "W" in self._done:
            return None
        # all_gather_into_tensor needs the same row count on every rank (advisor, round 5): checked once per batch size with a MIN / MAX
        # all-reduce — a collective, so every rank must get here with the same first-write state, which a synchronising step of ranks
        # that all accumulate alike guarantees (train_step's sync / loss_scale arguments are rank-uniform by contract).  A ragged
        # batch switches the exchange off for good: the stack then rides the all-reduce, which tolerates any per-rank batch.
        nb = int(gb.shape[0])
        ok = self._exchange_ok.get(nb)
        if ok is None:
            lohi = torch.tensor([nb, -nb], dtype=torch.int64, device=gb.device)
            torch.distributed.all_reduce(lohi, op=torch.distributed.ReduceOp.MIN, group=self.pg)
            ok = self._exchange_ok[nb] = bool(int(lohi[0]) == nb and int(lohi[1]) == -nb)
            if not ok:
                warnings.warn(f"E4TTrainer: per-rank batch sizes differ ({int(lohi[0])}..{-int(lohi[1])}): the head's factor exchange is off, its stacked gradient is all-reduced")
                self._exchange_ok = _AlwaysFalse()
        if not ok:
            return None
        w = gb.shape[1]
        mine = torch.cat([gb, Z], dim=1)
        rows = torch.empty((self.world * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=mine.device)
        if self.comm_timing is not None and mine.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.comm_timing.setdefault("enqueue", {})["W(factors)"] = ev
        if self._lib_comm is not None:
            self._lib_comm.all_gather_into_tensor(rows, mine).wait()      # the stream is ordered after it before `mine` can be recycled
        else:
            torch.distributed.all_gather_into_tensor(rows, mine, group=self.pg)
        self._done.add("W")
        self._factor_bytes = mine.numel() * mine.element_size()
        return rows[:, :w], rows[:, w:]

    def _take_head_factors(self, gb, Z):
        """_HeadFn.backward offers the factors of the stacked first_linears gradient (under data parallelism: every rank's rows, already
        gathered — region W is then marked reduced).  Taken only when optimizer_step follows this very backward and nothing else needs
        the stack: a synchronising step, no accumulated micro-batches, no gradient clip."""
        if not (self.factored_head_update and self._armed and not self._accum_pending and self.max_grad_norm is None and self._stack_shape):
            return False
        if self.world > 1 and "W" not in self._done:          # local factors only: the stack has to ride the all-reduce
            return False
        self._head_factors = (gb, Z)
        return True

    def _reduce_region(self, key, force=False):
        """enqueue the all-reduce of one region; during the backward only while a synchronising step is armed, `force` for the
        sweep after the backward that picks up whatever no hook announced"""
        if self.regions is None or key in self._done or not (self._armed or force):
            return
        self._done.add(key)
        a, b = self.regions[key]
        if b <= a:
            return
        if self.comm_timing is not None and self.flat.grad.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.comm_timing.setdefault("enqueue", {})[key] = ev
        g = self.flat.grad
        bucket = 64 << 20          # 256 MB fp32: xGMI rings are per-link bound, large buckets run them at rate
        for o in range(a, b, bucket):
            self._works.append((key, self._all_reduce(g[o:min(o + bucket, b)], async_op=True)))

    def _all_reduce(self, t, async_op=False):
        """SUM all-reduce of a slice of the flat gradient on the configured back end; async_op: a handle whose wait() orders the current
        stream after it"""
        if self._lib_comm is None:
            return torch.distributed.all_reduce(t, group=self.pg, async_op=async_op)
        h = self._lib_comm.all_reduce(t)
        if async_op:
            return h
        h.wait()
        return None

    def all_reduce_grads(self, defer=None):
        """Wait for the regions' all-reduces (enqueueing whatever no hook announced).  `defer` names ONE region whose handles are not
        waited for but returned, with its bounds: region D is final — and its all-reduce starts — only when the backward ends, so without a
        gradient clip the step runs AdamW on everything else under it and on D afterwards (optimizer_step(deferred=...))."""
        if not self._comm:
            return None
        if self.regions is None:
            g = self.flat.grad
            bucket = 64 << 20
            for o in range(0, g.numel(), bucket):
                self._all_reduce(g[o:o + bucket])
            return None
        for key in ("U", "W", "H", "D", "T"):      # whatever was not triggered during the backward
            if self.regions[key][1] > self.regions[key][0]:
                self._reduce_region(key, force=True)
        self._mark("wait_begin")
        late = [w for k, w in self._works if k == defer]
        for k, w in self._works:
            if k != defer:
                w.wait()
        self._mark("wait_end")
        self._works, self._done = [], set()
        return (self.regions[defer], late) if late else None

    def _mark(self, name):
        if self.comm_timing is not None and self.flat.grad.is_cuda:
            self.comm_timing[name] = torch.cuda.Event(enable_timing=True)
            self.comm_timing[name].record()

    def clip_grad_norm(self):
        """tuning_e4t.py:329-335 — global L2 norm over the flat gradient (one reduction kernel), scale folded on the device
        (no host sync).  Under DP the gradient is the SUM over ranks here, so the norm is taken of sum/world."""
        if self.max_grad_norm is None:
            return
        norm = ops.backend().sumsq(self.flat.grad).sqrt() / self.world
        self.flat.grad.mul_(torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0))

    def optimizer_step(self, deferred=None):
        """AdamW over the flat buffers.  deferred = ((lo, hi), handles) from all_reduce_grads(defer=...): the elements outside [lo, hi)
        are updated first, under that region's all-reduce, then the handles are waited for and [lo, hi) follows (element-wise update:
        the split changes no bit)."""
        if self._hyper is not None:
            # step-graph mode: lr / bias corrections / gradient scale live in device memory (refreshed by the host before every step, eager
            # or replayed), so the captured launch is the same launch at every step
            if not self._capturing:
                self.step_count += 1
                self._write_hyper()

            def adamw(lo, hi):
                ops.backend().adamw_hyper(self.flat.data[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self._hyper,
                                          self.betas[0], self.betas[1], self.eps, self.wd)
        else:
            self.step_count += 1

            def adamw(lo, hi):
                ops.backend().adamw(self.flat.data[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                                    self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0 / self.world)
        factors, self._head_factors = self._head_factors, None
        if factors is not None:
            # the stack leads the flat buffer: [0, w_end) is updated from the factors, the ordinary launch starts behind it
            n, rows, cols = self._stack_shape
            w_end = self._stack_grad_off + n * rows * cols
            assert self._stack_grad_off == 0 and (deferred is None or deferred[0][0] >= w_end)
            st = lambda buf: buf[:w_end].view(n, rows, cols)
            gb, Z = factors
            be = ops.backend()
            if self._hyper is not None:
                be.adamw_rank(st(self.flat.data), st(self.exp_avg), st(self.exp_avg_sq), gb, Z, 0.0, self.betas[0], self.betas[1], self.eps, self.wd, 0,
                              hyper=self._hyper)
            else:
                be.adamw_rank(st(self.flat.data), st(self.exp_avg), st(self.exp_avg_sq), gb, Z, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                              self.step_count, 1.0 / self.world)
            inner = adamw

            def adamw(lo, hi):
                if hi > max(lo, w_end):
                    inner(max(lo, w_end), hi)
        if deferred is None:
            adamw(0, self.flat.numel)
        else:
            (lo, hi), handles = deferred
            if lo > 0:
                adamw(0, lo)
            if hi < self.flat.numel:
                adamw(hi, self.flat.numel)
            self._mark("late_wait_begin")
            for w in handles:
                w.wait()
            self._mark("late_wait_end")
            adamw(lo, hi)
        ops.bump_weights_epoch()

    # ---- the whole step as ONE HIP graph ----------------------------------------------------------------------------------------
    # A step is ~2250 launches (SD-1.4) issued from Python at ~20 us each.  At B = 16 the GPU needs 100+ ms for them and the host keeps
    # ahead; at small batches (BASELINE configs[4] runs SD-2.x at B = 1 per GPU: 17 ms of kernels) the step is host-bound (48 ms).  Shapes
    # are static per (config, batch): the step — VAE encode, both UNet passes, encoder, text encoder, backward, clip, AdamW, zero-grad — is
    # captured once per input signature with torch.cuda.graph and replayed; inputs go through static buffers, the random draws are
    # torch's graph-safe Philox draws, AdamW's step-dependent scalars are read from device memory (e4t_adamw_hyper).  The first step of a
    # signature runs eagerly (it is also the warm-up every lazy initialisation needs), the second one captures and replays.
    # Under a communicator (round 5: BASELINE configs[4] IS one image per GPU on eight GPUs) the gradient regions' all-reduces are captured
    # WITH the step: the hooks run during the capture, RCCL's enqueues on its own stream become graph nodes joined to the step's stream by
    # the handles' waits, and every rank captures the same sequence.  The replica check (a host read) runs outside the replay.  Should a
    # capture fail (an RCCL / HIP build that refuses a collective under capture), the trainer says so once and goes on eagerly.
    def enable_step_graph(self, on=True):
        if on and (not self.flat.data.is_cuda or self.text_trainable or self._graph_failed):
            return False
        if on and self._lib_comm is not None:
            return False          # the library's collective stream is not a torch stream: its work is not captured with the step
        self._step_graph_on = bool(on)
        if on and self._hyper is None:
            # allocated once and kept for the trainer's lifetime: captured graphs hold its raw pointer
            self._hyper = torch.zeros(4, dtype=f32, device=self.device)
            self._hyper_ring = [[torch.zeros(4, dtype=f32).pin_memory(), None] for _ in range(self._HYPER_SLOTS)]
        if not on:
            # graphs own whole-step memory pools: drop them (a later enable re-captures); _hyper stays
            self._step_graphs, self._seen_sigs = {}, set()
        return self._step_graph_on

    _HYPER_SLOTS = 8

    def _write_hyper(self):
        """AdamW's step-dependent scalars -> device memory, ordered on the step's stream.  With graph replay the host enqueues a step in
        ~1 ms against 40-100 ms of GPU work and runs several steps ahead, so ONE pinned staging buffer would be overwritten before
        its queued copy executed (round-4 advisor finding: later steps' lr / bias corrections seen by earlier updates).  The staging is a
        ring of pinned slots, each with the event of the copy that last read it; a slot is rewritten only after that copy has run,
        which also bounds the host's lead to _HYPER_SLOTS steps."""
        import ctypes
        libm = getattr(E4TTrainer, "_libm", None)
        if libm is None:
            libm = E4TTrainer._libm = ctypes.CDLL("libm.so.6")
            libm.powf.restype, libm.powf.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
            libm.sqrtf.restype, libm.sqrtf.argtypes = ctypes.c_float, [ctypes.c_float]
        f = lambda x: ctypes.c_float(x).value
        t = float(self.step_count)
        slot = self._hyper_ring[self._hyper_i % len(self._hyper_ring)]
        self._hyper_i += 1
        host, ev = slot
        if ev is not None:
            ev.synchronize()
        # exactly e4t_adamw's host arithmetic (fp32 powf / sqrtf): 1 - beta1^t, sqrt(1 - beta2^t)
        host[0] = self.lr
        host[1] = f(1.0 - libm.powf(self.betas[0], t))
        host[2] = libm.sqrtf(f(1.0 - libm.powf(self.betas[1], t)))
        host[3] = 1.0 / self.world
        self._hyper.copy_(host, non_blocking=True)
        ev = slot[1] = torch.cuda.Event()
        ev.record()

    def _graphed_step(self, pixel_values, input_ids, placeholder_idx, noise, timesteps, vae_eps, latents):
        ins = dict(pixel_values=pixel_values, input_ids=input_ids, placeholder_idx=placeholder_idx, noise=noise, timesteps=timesteps,
                   vae_eps=vae_eps, latents=latents)
        sig = tuple((k, tuple(v.shape), v.dtype) for k, v in ins.items() if v is not None)
        ent = self._step_graphs.get(sig)
        if ent is None:
            if sig not in self._seen_sigs:            # first step of this signature: eager (= the warm-up)
                self._seen_sigs.add(sig)
                return self._train_step(**ins)
            static = {k: v.clone() for k, v in ins.items() if v is not None}
            g = torch.cuda.CUDAGraph()
            self._capturing = True
            captured = False
            try:
                with ops.capture_guard():
                    torch.cuda.synchronize()
                    if self._comm:
                        # ProcessGroupNCCL's watchdog thread polls the events of the eager steps' collectives (hipEventQuery every ~100 ms):
                        # under the default GLOBAL capture mode such a call from another thread invalidates the capture or kills the watchdog
                        # ("operation not permitted when stream is capturing", 4 of 12 runs of tests/rccl_one_rank.py).  Give it time to
                        # retire the finished work, and capture thread-locally so that its remaining calls are none of the capture's business.
                        # (The device is already quiescent — synchronize() above; what the pause covers is the watchdog's own polling
                        # loop, which ProcessGroupNCCL offers no call to drain.  thread_local mode is the guard, the pause only makes
                        # its warning-free path the common one; a capture that still fails is handled collectively below.)
                        time.sleep(0.5)
                    with torch.cuda.graph(g, capture_error_mode="thread_local" if self._comm else "global"):
                        out = self._train_step(**{k: static.get(k) for k in ins})
                captured = True
            except Exception as e:                    # e.g. a collective the RCCL build will not capture: eager from here on
                if not self._comm:
                    raise
                warnings.warn(f"E4TTrainer: capturing the step with its collectives failed ({type(e).__name__}: {e}); running eagerly")
                self._works, self._done = [], set()
                torch.cuda.synchronize()
            finally:
                self._capturing = False
            if self._comm:
                # the decision is COLLECTIVE (advisor, round 5): a rank that fell back alone would enqueue its step from Python while
                # the others replay theirs in ~1 ms and then wait for it in every all-reduce.  One MIN all-reduce of "captured":
                # either every rank replays or every rank drops its graph and runs eagerly from here on.
                flag = torch.tensor([1 if captured else 0], dtype=torch.int32, device=self.device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.pg)
                if not bool(int(flag)):
                    if captured:
                        warnings.warn("E4TTrainer: another rank could not capture the step; running eagerly on every rank")
                    del g
                    self._step_graph_on, self._graph_failed = False, True
                    self._works, self._done = [], set()
                    return self._train_step(**ins)
            ent = self._step_graphs[sig] = (g, static, out)
        g, static, out = ent
        for k, v in static.items():
            if ins[k].data_ptr() != v.data_ptr():
                v.copy_(ins[k], non_blocking=True)
        self.step_count += 1
        self._write_hyper()
        g.replay()
        ops.bump_weights_epoch()
        if self.world > 1 and self.replica_check_every > 0 and self.step_count % self.replica_check_every == 0:
            self.check_replicas()
        return tuple(o.clone() for o in out)

    def zero_grad(self):
        enc = self.encoder
        gW = getattr(enc, "_gW", None)
        ours = self._stack_shape is not None and gW is not None and gW.data_ptr() == self.flat.grad.data_ptr() + 4 * self._stack_grad_off
        if ours and getattr(enc, "_stack_grad_is_zero", False) and self._stack_grad_off == 0:
            # nobody wrote the stack since it was last cleared (factored update): 845 MB less to clear.  Invariant: between zero_grad and
            # the head's backward only _HeadFn.backward writes flat.grad's stack region, and it lowers the mark when it does.
            n, rows, cols = self._stack_shape
            self.flat.grad[n * rows * cols:].zero_()
        else:
            self.flat.grad.zero_()
        if self._stack_shape is not None:
            enc._stack_grad_is_zero = ours
        self._accum_pending = False

    # ---- training state (accelerator.save_state / load_state of the reference, pretrain_e4t.py:536-558,659-663) ----------
    def state_dict(self):
        return dict(params=self.flat.data.detach().cpu().clone(), exp_avg=self.exp_avg.cpu().clone(), exp_avg_sq=self.exp_avg_sq.cpu().clone(),
                    step_count=self.step_count, lr=self.lr)

    def load_state_dict(self, sd):
        if sd["params"].numel() != self.flat.data.numel():
            raise RuntimeError(f"training state holds {sd['params'].numel()} parameters, this trainer {self.flat.data.numel()}")
        self.flat.data.copy_(sd["params"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step_count"])
        self.flat.grad.zero_()
        self.sync_replicas(include_moments=True)  # a per-rank-different checkpoint read must not start diverged replicas
        ops.bump_weights_epoch()                 # bf16 compute copies of the trainable weights are stale now

    def train_step(self, pixel_values, input_ids, placeholder_idx, noise=None, timesteps=None, vae_eps=None, latents=None,
                   sync=True, loss_scale=1.0):
        """One training step (see _train_step); replayed from the step's HIP graph when enable_step_graph() is on and the call is a plain
        synchronising step."""
        def run():
            # (a replayed graph bakes in the first-write / factored treatment of the head's gradient stack: only from a clean gradient)
            if self._step_graph_on and sync and loss_scale == 1.0 and self._next_px is None and not self._pref and not self._accum_pending:
                return self._graphed_step(pixel_values, input_ids, placeholder_idx, noise, timesteps, vae_eps, latents)
            return self._train_step(pixel_values, input_ids, placeholder_idx, noise, timesteps, vae_eps, latents, sync, loss_scale)
        ts = self._training_stream()
        if ts is None or self._step_graph_on:          # (a replayed graph's branches run on the graph executor's own streams, wherever it is launched)
            return run()
        # data parallel, and the caller's stream shares RCCL's hardware queue: the step runs on a stream of the trainer's that does not
        cur = torch.cuda.current_stream(self.device)
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            out = run()
        cur.wait_stream(ts)
        for t in out:
            t.record_stream(cur)
        return out

    def _training_stream(self):
        """Under a communicator on a GPU: the stream the step must run on for its all-reduces to OVERLAP with it, or None (the caller's).
        ROCm maps streams onto a handful of hardware queues and two streams on one queue serialise (_new_side_stream); RCCL launches on a
        stream of ProcessGroupNCCL's choosing, and in a fresh process that stream shared the default stream's queue (tools/probe/
        stream_queues.py) — every region's all-reduce would then sit IN the backward's kernel sequence instead of beside it.  Chosen
        once, at the first step, by probing: a spin kernel on the candidate, a small all-gather issued from an idle stream, taken when the
        collective completes while the spin still runs.  The probe is a collective, so every rank probes every candidate (the caller's
        stream, then five new ones) whatever it found; which candidate a rank takes is its own business."""
        if self._train_stream_probed:
            return self._train_stream
        self._train_stream_probed = True
        if not self._comm or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return None
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        cands = [cur] + [torch.cuda.Stream(device=dev) for _ in range(5)]
        if self._lib_comm is not None:
            # the library's collective stream is known: ask the hardware-queue question directly, no collective involved
            ext = torch.cuda.ExternalStream(self._lib_comm.stream_handle(), device=dev)
            found = [_runs_beside(c, ext, dev) for c in cands]
            self.stream_probe = found
            self._train_stream = None if (found[0] or not any(found)) else cands[found.index(True)]
            return self._train_stream
        src = torch.ones(1024, device=dev)
        dst = torch.empty(self.world * 1024, device=dev)
        torch.distributed.all_gather_into_tensor(dst, src, group=self.pg)          # the communicator's lazy set-up is not part of the probe
        found = []
        for c in cands:
            idle = next((x for x in cands if x is not c and _runs_beside(c, x, dev)), cands[-1] if c is not cands[-1] else cands[0])
            torch.distributed.barrier(group=self.pg)
            torch.cuda.synchronize(dev)
            e = torch.cuda.Event()
            with torch.cuda.stream(c):
                torch.cuda._sleep(60_000_000)          # ~29 ms (20 M cycles = 9.6 ms here): room for the ranks' skew behind the barrier
                e.record(c)
            with torch.cuda.stream(idle):
                w = torch.distributed.all_gather_into_tensor(dst, src, group=self.pg, async_op=True)
            ok = False
            while not e.query():
                if w.is_completed():
                    ok = True
                    break
            w.wait()
            torch.cuda.synchronize(dev)
            found.append(ok)
        self.stream_probe = found          # diagnostics: per candidate, did a collective run beside it (index 0 = the caller's stream)
        if found[0] or not any(found):
            if not any(found):
                warnings.warn("E4TTrainer: no stream on which RCCL's collectives overlap with the step was found; the all-reduces will serialise with the backward")
            return None
        self._train_stream = cands[found.index(True)]
        return self._train_stream

    def _train_step(self, pixel_values, input_ids, placeholder_idx, noise=None, timesteps=None, vae_eps=None, latents=None,
                    sync=True, loss_scale=1.0):
        """Full step.  Random draws may be passed in (parity tests) or are sampled on the device.
        Gradient accumulation (``accelerator.accumulate``, pretrain_e4t.py:595): call with ``sync=False`` and
        ``loss_scale=1/k`` for the first k-1 micro-batches — gradients accumulate locally, no collective, no optimiser step —
        and with ``sync=True`` (same loss_scale) for the k-th."""
        dev = self.device
        B = pixel_values.shape[0]
        pre_vision, pre_latents, vis_ev = self._take_prefetched(pixel_values) if self._pref else (None, None, None)
        if pre_vision is not None:
            self._vision, self._vision_event = pre_vision, vis_ev
        if latents is None and pre_latents is not None and vae_eps is None:
            latents = pre_latents
        if latents is None:
            hl, wl = pixel_values.shape[2] // 8, pixel_values.shape[3] // 8
            if vae_eps is None:
                vae_eps = torch.randn((B, 4, hl, wl), device=dev)
            latents = self.encode_latents(pixel_values, vae_eps)
        if noise is None:
            noise = torch.randn_like(latents)
        if timesteps is None:
            timesteps = torch.randint(0, self.acp.shape[0], (B,), device=dev).long()
        loss, loss_diff, loss_reg = self.losses(pixel_values, latents, noise, timesteps, input_ids, placeholder_idx)
        if self._next_px is not None:        # announced by prefetch(): the next batch's frozen encoders start with this backward
            self._start_prefetch()           # (starting them with the step instead: 101.95 vs 101.8 ms, profiles/r05_ab/r05b_at_step.json)
        self._armed = bool(sync)             # micro-batches that only accumulate start no collectives
        self._up_events = 0
        if self.comm_timing is not None and self.flat.grad.is_cuda:
            self.comm_timing.clear()
            self.comm_timing["backward_begin"] = torch.cuda.Event(enable_timing=True)
            self.comm_timing["backward_begin"].record()
        Fn.set_inplace_param_grads(True)     # weight / bias gradients accumulate straight into the flat buffer (functional.py)
        try:
            (loss if loss_scale == 1.0 else loss * loss_scale).backward()
        finally:
            Fn.set_inplace_param_grads(False)
        self._armed = False
        if not sync:
            self._accum_pending = True
            return loss.detach(), loss_diff.detach(), loss_reg.detach()
        # (the gradient clip needs the norm of the WHOLE reduced gradient: with it every region is waited for first)
        late = self.all_reduce_grads(defer="D" if self.max_grad_norm is None else None)
        self.deferred_region = "D" if late else None          # diagnostics / tests
        self.clip_grad_norm()
        self.optimizer_step(late)
        self.zero_grad()
        if self.world > 1 and self.replica_check_every > 0 and self.step_count % self.replica_check_every == 0 and not self._capturing:
            self.check_replicas()
        return loss.detach(), loss_diff.detach(), loss_reg.detach()

    def comm_report(self):
        """ms since the start of the backward at which each gradient region's all-reduce was enqueued in the last synchronising
        step, and the time the step's stream then spent waiting for the collectives ("exposed"): needs comm_timing = {} beforehand"""
        t = self.comm_timing
        if not t or "wait_end" not in t:
            return None
        torch.cuda.synchronize()
        b = t["backward_begin"]
        return dict(enqueue_ms_after_backward_start={k: b.elapsed_time(e) for k, e in t.get("enqueue", {}).items()},
                    wait_begin_ms=b.elapsed_time(t["wait_begin"]),
                    # the step's stream waiting for collectives: before AdamW, plus — when region D was deferred — between AdamW of the rest and of D
                    exposed_wait_ms=t["wait_begin"].elapsed_time(t["wait_end"]) + (t["late_wait_begin"].elapsed_time(t["late_wait_end"]) if "late_wait_end" in t else 0.0),
                    deferred_region="D" if "late_wait_end" in t else None,
                    region_bytes={k: 4 * (hi - lo) for k, (lo, hi) in (self.regions or {}).items()},
                    head_factor_bytes_per_rank=getattr(self, "_factor_bytes", 0),     # > 0: region W was exchanged as factors, not all-reduced
                    # _training_stream(): did a collective overlap with [the caller's stream, five new ones]; did the step move off the caller's
                    collectives_run_beside=getattr(self, "stream_probe", None), step_on_own_stream=self._train_stream is not None)
