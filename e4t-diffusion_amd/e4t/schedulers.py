"""Samplers for the inference row.

Reference: ``inference.py:60-67,113`` maps ``--scheduler_type`` to diffusers schedulers (default "ddim") and the
in-training sampler builds a ``DDIMScheduler`` (``pretrain_e4t.py:450``); diffusers itself is a third-party dependency that
is not under /root/reference and not installed (requirements.txt pins diffusers==0.14.0).  ``DDIMScheduler`` below follows
that version's published algorithm and public surface (``set_timesteps``, ``timesteps``, ``scale_model_input``,
``init_noise_sigma``, ``order``, ``step(...).prev_sample``, ``from_config`` / ``from_pretrained`` of a
``scheduler_config.json``).  The other five names of the reference's table are not built.

MI355X-first detail: without sample clipping a DDIM update is linear in (sample, model_output, noise),
    prev = c_sample * sample + c_pred * model_output + c_noise * noise,
for both epsilon and v prediction.  ``coefficients(i, eta)`` exposes the three numbers so the pipeline can run guidance
and the update as ONE kernel (``e4t_guided_step``) whose coefficients live in device memory — a captured hipGraph of the
whole denoising step is then replayed for every timestep.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", **unused):
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__.__name__}")
        if prediction_type not in ("epsilon", "v_prediction", "sample"):
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                           clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self._acp = [float(a) for a in self.alphas_cumprod]              # fp32 values, as python floats
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    # ---- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def stable_diffusion(cls, prediction_type="epsilon"):
        """the scheduler_config.json shipped with the Stable Diffusion 1.x / 2.x checkpoints"""
        return cls(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                   steps_offset=1, prediction_type=prediction_type)

    @classmethod
    def from_config(cls, config: dict):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        f = os.path.join(path, subfolder or "", "scheduler_config.json")
        with open(f) as fh:
            return cls.from_config(json.load(fh))

    # ---- schedule ---------------------------------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config["num_train_timesteps"]
        if num_inference_steps > n_train:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `num_train_timesteps`: {n_train}")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        ts = [i * ratio + self.config["steps_offset"] for i in range(num_inference_steps)][::-1]
        self.timesteps = torch.tensor(ts, dtype=torch.int64, device=device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, timestep: int):
        prev = timestep - self.config["num_train_timesteps"] // self.num_inference_steps
        return self._acp[timestep], (self._acp[prev] if prev >= 0 else self.final_alpha_cumprod)

    def coefficients(self, timestep: int, eta: float = 0.0):
        """(c_sample, c_pred, c_noise) of the linear update at `timestep` (requires clip_sample == False)"""
        if self.config["clip_sample"] or self.config["prediction_type"] == "sample":
            raise ValueError("the update is linear only without sample clipping and for epsilon / v prediction")
        a_t, a_prev = self._alphas(int(timestep))
        sa, sb, sap = math.sqrt(a_t), math.sqrt(1.0 - a_t), math.sqrt(a_prev)
        var = (1.0 - a_prev) / (1.0 - a_t) * (1.0 - a_t / a_prev)
        std = eta * math.sqrt(max(var, 0.0))
        d = math.sqrt(max(1.0 - a_prev - std * std, 0.0))
        if self.config["prediction_type"] == "epsilon":
            return sap / sa, d - sap * sb / sa, std
        return sap * sa + d * sb, d * sa - sap * sb, std

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False, generator=None,
             variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t = int(timestep)
        a_t, a_prev = self._alphas(t)
        if eta > 0 and variance_noise is None:
            variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        linear = not self.config["clip_sample"] and self.config["prediction_type"] != "sample" and not use_clipped_model_output
        if linear and sample.dtype == torch.float32 and model_output.dtype == torch.float32:
            cs, cp, cn = self.coefficients(t, eta)
            coef = torch.tensor([0.0, cs, cp, cn], dtype=torch.float32, device=sample.device)
            prev = ops.backend().guided_step(model_output.contiguous(), sample.contiguous(), coef,
                                             noise=variance_noise.contiguous() if eta > 0 else None, cfg=False)
            x0 = None
        else:                                   # clipped / "sample" prediction: the generic formulation, elementwise torch
            b_t = 1.0 - a_t
            pt = self.config["prediction_type"]
            if pt == "epsilon":
                x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            elif pt == "sample":
                x0 = model_output
                model_output = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
            else:
                x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
                model_output = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
            if self.config["clip_sample"]:
                x0 = x0.clamp(-1, 1)
            var = (1.0 - a_prev) / (1.0 - a_t) * (1.0 - a_t / a_prev)
            std = eta * var ** 0.5
            if use_clipped_model_output:
                model_output = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
            prev = a_prev ** 0.5 * x0 + (1.0 - a_prev - std ** 2) ** 0.5 * model_output
            if eta > 0:
                prev = prev + std * variance_noise
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        a = acp[timesteps].sqrt().view(-1, *([1] * (original_samples.dim() - 1)))
        s = (1 - acp[timesteps]).sqrt().view(-1, *([1] * (original_samples.dim() - 1)))
        return a * original_samples + s * noise

    def __len__(self):
        return self.config["num_train_timesteps"]


SCHEDULER_MAPPING = {"ddim": DDIMScheduler}            # inference.py:60-67 (the other five entries are not built)
