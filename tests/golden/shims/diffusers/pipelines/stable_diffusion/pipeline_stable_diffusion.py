"""The parts of diffusers 0.14's StableDiffusionPipeline that the reference's subclass relies on
(e4t/pipeline_stable_diffusion_e4t.py: __init__ :43, register_modules :60, check_inputs :125, _execution_device :137,
prepare_latents :166, prepare_extra_step_kwargs :176, progress_bar :180, decode_latents / run_safety_checker / numpy_to_pil)."""
import contextlib
import inspect
from dataclasses import dataclass
from typing import Any, Optional

import torch

from ...utils import BaseOutput


@dataclass
class StableDiffusionPipelineOutput(BaseOutput):
    images: Any = None
    nsfw_content_detected: Optional[Any] = None


class _Bar:
    def update(self, n=1):
        pass


class StableDiffusionPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker, feature_extractor, requires_safety_checker=True):
        self.register_modules(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler,
                              safety_checker=safety_checker, feature_extractor=feature_extractor)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return next(self.unet.parameters()).device

    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        accepted = set(inspect.signature(self.scheduler.step).parameters.keys())
        extra = {}
        if "eta" in accepted:
            extra["eta"] = eta
        if "generator" in accepted:
            extra["generator"] = generator
        return extra

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        yield _Bar()

    def decode_latents(self, latents):
        image = self.vae.decode(latents / self.vae.config.scaling_factor).sample
        return (image / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()

    def run_safety_checker(self, image, device, dtype):
        return image, None

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        return [Image.fromarray(i) for i in (images * 255).round().astype("uint8")]
