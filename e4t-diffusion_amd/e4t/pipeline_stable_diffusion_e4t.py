"""E4T sampling pipeline — same class name, constructor and ``__call__`` arguments as the reference's
``StableDiffusionE4TPipeline`` (e4t/pipeline_stable_diffusion_e4t.py:30-250), on the native modules.

What one denoising step does is the reference's (``:178-214``): UNet encoder pass on the current latents with the ""
context -> E4T encoder -> domain embedding written into the prompt's token embeddings -> text encoder -> UNet on
[latents, latents] with [""-context, prompt-context] -> guidance -> scheduler update.  What is organised differently for
the MI355X:

  * everything that does not depend on the latents is evaluated once per call, not once per step: the CLIP-ViT features of
    the conditioning image (``E4TEncoder.encode_vision``), the ""-prompt context, the weight-offset ``W_eff`` (cached by
    the bank while the parameters do not change);
  * guidance + the DDIM update are one kernel (``e4t_guided_step``) reading the UNet's NHWC output in place;
  * at 1-2 images a step is ~2000 short kernel launches; the whole step is captured once into a hipGraph
    (``torch.cuda.CUDAGraph``) and replayed per timestep: the timestep and the update coefficients are read from small
    device buffers that are refreshed between replays.  ``use_graph=None`` -> on for <= 2 images per call when the
    scheduler update is linear and eta == 0 (measured: 13.6 -> 11.8 ms/step at one image, nothing from four images up,
    where the step is GPU-bound); ``use_graph=False`` -> eager.

The tokenizer is whatever object the caller passes (``transformers.CLIPTokenizer`` in inference.py): it needs
``__call__(text, padding=, truncation=, max_length=, return_tensors="pt", add_special_tokens=)``, ``add_tokens``,
``convert_tokens_to_ids``, ``model_max_length`` and ``__len__``.  The safety checker is not built (the reference runs the
pipeline with ``safety_checker=None``, inference.py:117-119).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Union

import numpy as np
import torch

from . import ops
from .schedulers import DDIMScheduler


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]] = None


def preprocess(image):
    """PIL image(s) / tensor -> float32 (N,3,H,W) in [-1,1]  (pipeline_stable_diffusion_e4t.py:12-27)"""
    if isinstance(image, torch.Tensor):
        return image
    if not isinstance(image, (list, tuple)):
        image = [image]
    if isinstance(image[0], torch.Tensor):
        return torch.cat(list(image), dim=0)
    arr = np.concatenate([np.array(i)[None, :] for i in image], axis=0).astype(np.float32) / 255.0
    return torch.from_numpy(2.0 * arr.transpose(0, 3, 1, 2) - 1.0)


class StableDiffusionE4TPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, e4t_encoder, scheduler, safety_checker=None, feature_extractor=None,
                 e4t_config=None, requires_safety_checker: bool = True, already_added_placeholder_token: bool = False):
        if safety_checker is not None:
            raise NotImplementedError("the safety checker is not part of this build: pass safety_checker=None")
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.e4t_encoder = e4t_encoder
        self.safety_checker, self.feature_extractor = None, feature_extractor
        boc = getattr(vae, "block_out_channels", None) or getattr(getattr(vae, "config", None), "block_out_channels", (0,) * 4)
        self.vae_scale_factor = 2 ** (len(boc) - 1)
        cfg = e4t_config if not isinstance(e4t_config, dict) else _Attr(e4t_config)
        if not already_added_placeholder_token:                                       # reference :45-53
            if self.tokenizer.add_tokens(cfg.placeholder_token) == 0:
                raise ValueError(f"The tokenizer already contains the token {cfg.placeholder_token}. Please pass a different "
                                 "`placeholder_token` that is not already in the tokenizer.")
            text_encoder.resize_token_embeddings(len(tokenizer))
        self.placeholder_token = cfg.placeholder_token
        self.placeholder_token_id = tokenizer.convert_tokens_to_ids(cfg.placeholder_token)
        ids = self.tokenizer(cfg.domain_class_token, add_special_tokens=False, return_tensors="pt").input_ids[0]
        assert ids.size(0) == 1, "the domain class must be a single token"
        emb = text_encoder.get_input_embeddings()
        self.class_embed = emb(ids.to(emb.weight.device)).detach()                    # (1, d)   reference :55-59
        self.domain_embed_scale = cfg.domain_embed_scale
        self._progress = {}
        self._graph = None

    # ---- housekeeping the reference inherits from DiffusionPipeline -------------------------------------------------
    @property
    def device(self):
        return next(self.unet.parameters()).device

    _execution_device = device

    def to(self, device):
        for m in (self.vae, self.text_encoder, self.unet, self.e4t_encoder):
            m.to(device)
        self.class_embed = self.class_embed.to(device)
        self._graph = None
        return self

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        self.unet.enable_xformers_memory_efficient_attention()          # selects the native fused-attention processor

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    # ---- reference helpers ----------------------------------------------------------------------------------------
    def prepare_for_e4t(self, prompt, device):
        """reference :68-88"""
        tok = self.tokenizer
        kw = dict(padding="max_length", truncation=True, max_length=tok.model_max_length, return_tensors="pt")
        ids_e = tok("", **kw).input_ids
        ids = tok(prompt, **kw).input_ids
        try:
            idx = ids[0].tolist().index(self.placeholder_token_id)
        except ValueError:
            raise ValueError(f"Your prompt may not have the placeholder_token={self.placeholder_token}")
        with torch.no_grad():
            ctx_e = self.text_encoder(ids_e.to(device))[0]
            emb = self.text_encoder.get_input_embeddings()(ids.to(device))
        return dict(placeholder_token_id_idx=idx, encoder_hidden_states_for_e4t=ctx_e, inputs_embeds=emb)

    def check_inputs(self, prompt, height, width, callback_steps):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is None or not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {batch_size}.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype).to(device) for g in generator])
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents):
        return self.vae.decode_latents(latents).cpu().float().numpy()                # (B, H, W, 3) in [0, 1]

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        return [Image.fromarray(i) for i in (images * 255).round().astype("uint8")]

    # ---- one denoising step -----------------------------------------------------------------------------------------
    def _model_step(self, latents, t, s):
        """eps (or v) prediction for guidance: returns the UNet output tensor of the [uncond | cond] (or cond-only) batch"""
        bsz = latents.shape[0]
        x_in = self.scheduler.scale_model_input(latents, t)
        ctx_e = s["ctx_e"].expand(bsz, -1, -1)
        enc = self.unet(x_in, t, ctx_e, return_encoder_outputs=True)                                   # :189
        if s["vision"] is not None:
            domain = self.e4t_encoder(x=s["pixels"], unet_down_block_samples=enc["down_block_samples"], vision=s["vision"])
        else:
            domain = self.e4t_encoder(x=s["pixels"], unet_down_block_samples=enc["down_block_samples"])  # :192
        domain = self.class_embed.expand(bsz, -1).to(domain.dtype) + s["scale"] * domain              # :194
        emb = s["inputs_embeds"].expand(bsz, -1, -1).clone()
        emb[:, s["idx"], :] = domain.to(emb.dtype)                                                       # :195-196
        ctx = self.text_encoder(inputs_embeds=emb)[0].to(ctx_e.dtype)                                    # :198
        if s["cfg"]:
            return self.unet(torch.cat([x_in] * 2), t, torch.cat([ctx_e, ctx])).sample                 # :199-206
        return self.unet(x_in, t, ctx).sample

    def _fused_update(self, pred, latents, coef, cfg):
        """guidance + linear scheduler update in one kernel, in place on `latents`"""
        nhwc = pred.permute(0, 2, 3, 1)
        if nhwc.is_contiguous():                         # the native UNet hands out an NCHW view of NHWC storage
            ops.backend().guided_step(nhwc, latents, coef, cfg=cfg, pred_nhwc=True, out=latents)
        else:
            ops.backend().guided_step(pred.contiguous(), latents, coef, cfg=cfg, pred_nhwc=False, out=latents)

    # ---- the call -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, callback: Optional[Callable] = None, callback_steps: int = 1,
                 cross_attention_kwargs=None, image=None, domain_embed_scale: Optional[float] = None, use_graph: Optional[bool] = None):
        scale = self.domain_embed_scale if domain_embed_scale is None else domain_embed_scale
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        assert negative_prompt is None, "negative_prompt is not supported"           # reference :151
        assert prompt_embeds is None and negative_prompt_embeds is None and not cross_attention_kwargs
        if isinstance(prompt, list):
            if len(prompt) != 1:
                raise ValueError("one prompt per call (the reference reads the placeholder position of prompt[0] only, :76-79)")
            prompt = prompt[0]
        device = self.device
        cfg = guidance_scale > 1.0
        pixels = preprocess(image).to(device=device, dtype=torch.float32)
        if pixels.shape[0] != 1:
            raise ValueError("one conditioning image per call")
        e4t = self.prepare_for_e4t(prompt, device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        bsz = num_images_per_prompt
        latents = self.prepare_latents(bsz, self.unet.in_channels, height, width, torch.float32, device, generator, latents).contiguous()

        vision = None
        if hasattr(self.e4t_encoder, "encode_vision"):                                # image features: once, not per step
            cls, tokens = self.e4t_encoder.encode_vision(pixels)
            vision = (cls.expand(bsz, -1), tokens.expand(bsz, -1, -1))
        s = dict(ctx_e=e4t["encoder_hidden_states_for_e4t"], inputs_embeds=e4t["inputs_embeds"], idx=e4t["placeholder_token_id_idx"],
                 pixels=pixels.expand(bsz, -1, -1, -1), vision=vision, scale=scale, cfg=cfg)

        sch = self.scheduler
        fused = isinstance(sch, DDIMScheduler) and not sch.config["clip_sample"] and sch.config["prediction_type"] != "sample" and eta == 0.0
        if use_graph is None:       # measured (SD-1.4, 512 px, 50 steps): 13.6 -> 11.8 ms/step at 1 image, no gain from 4 images up
            use_graph = fused and device.type == "cuda" and bsz <= 2
        if use_graph and not fused:
            raise ValueError("graph replay needs the fused linear update (DDIMScheduler without sample clipping, eta == 0)")

        if fused:
            table = torch.tensor([[guidance_scale, *sch.coefficients(int(t), 0.0)] for t in timesteps.tolist()], dtype=torch.float32, device=device)
            coef = torch.empty(4, dtype=torch.float32, device=device)
            t_buf = torch.empty(1, dtype=torch.int64, device=device)

            def one_step():
                self._fused_update(self._model_step(latents, t_buf, s), latents, coef, cfg)

            graph = None
            for i in range(len(timesteps)):
                t_buf.copy_(timesteps[i:i + 1])
                coef.copy_(table[i])
                if not use_graph:
                    one_step()
                elif graph is None:
                    graph = self._capture(one_step, latents)
                    graph.replay()
                else:
                    graph.replay()
                if callback is not None and i % callback_steps == 0:
                    callback(i, timesteps[i], latents)
        else:
            import inspect
            accepted = set(inspect.signature(sch.step).parameters)          # prepare_extra_step_kwargs of the reference's base class
            extra = {}
            if "eta" in accepted:
                extra["eta"] = eta
            if "generator" in accepted and generator is not None:
                extra["generator"] = generator
            for i, t in enumerate(timesteps):
                pred = self._model_step(latents, t, s)
                if cfg:
                    u, c = pred.chunk(2)
                    pred = u + guidance_scale * (c - u)                              # :209-211
                latents = sch.step(pred.float(), t, latents, **extra).prev_sample    # :214
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)

        if output_type == "latent":
            images = latents
        else:
            images = self.decode_latents(latents)
            if output_type == "pil":
                images = self.numpy_to_pil(images)
        if not return_dict:
            return (images, None)
        return StableDiffusionPipelineOutput(images=images, nsfw_content_detected=None)

    def _capture(self, fn, latents):
        """Warm the step up on a side stream (weight copies, workspaces and allocator pools reach their steady state), restore
        the latents, then record the step into a graph on that stream."""
        keep = latents.clone()
        coefs_stream = torch.cuda.Stream(device=latents.device)
        coefs_stream.wait_stream(torch.cuda.current_stream(latents.device))
        with torch.cuda.stream(coefs_stream):
            fn()
            latents.copy_(keep)
        torch.cuda.current_stream(latents.device).wait_stream(coefs_stream)
        graph = torch.cuda.CUDAGraph()
        with ops.capture_guard(), torch.cuda.graph(graph, stream=coefs_stream):
            fn()
        latents.copy_(keep)               # capture does not execute: the first replay starts from the same state
        return graph


class _Attr(dict):
    __getattr__ = dict.__getitem__
