"""Samplers for the inference row.

Reference: ``inference.py:60-67,113`` maps ``--scheduler_type`` to diffusers schedulers (default "ddim") and the
in-training sampler builds a ``DDIMScheduler`` (``pretrain_e4t.py:450``); diffusers itself is a third-party dependency that
is not under /root/reference and not installed (requirements.txt pins diffusers==0.14.0).  ``DDIMScheduler`` below follows
that version's published algorithm and public surface (``set_timesteps``, ``timesteps``, ``scale_model_input``,
``init_noise_sigma``, ``order``, ``step(...).prev_sample``, ``from_config`` / ``from_pretrained`` of a
``scheduler_config.json``).  The other five names of the reference's table follow further down (generic update path).

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



# ----------------------------------------------------------------------------------------------------------------------
# The other five samplers of inference.py:60-67, restated from diffusers 0.14 (same defaults as the Stable Diffusion
# scheduler configs).  They run through the pipeline's generic path (scale_model_input / step(...).prev_sample, elementwise
# torch ops on the (B,4,h,w) latents — a few microseconds per step next to the UNet); only DDIM has the fused, graph-replayed
# update.  Parity unpinned (diffusers is not installed): tests/test_schedulers.py checks each of them against the closed-form
# probability-flow solution of a Gaussian toy problem.
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 prediction_type="epsilon", **extra):
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__.__name__}")
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                           prediction_type=prediction_type, **extra)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_inference_steps = None

    @classmethod
    def stable_diffusion(cls, prediction_type="epsilon", **kw):
        return cls(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type=prediction_type, **kw)

    @classmethod
    def from_config(cls, config: dict):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        with open(os.path.join(path, subfolder or "", "scheduler_config.json")) as fh:
            return cls.from_config(json.load(fh))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config["num_train_timesteps"]


class _SigmaBase(_Base):
    """k-diffusion parameterisation shared by the Euler / Euler-ancestral / LMS samplers: sigma = sqrt((1 - abar) / abar),
    timesteps = linspace(0, T-1, n) reversed (fractional), sigmas interpolated at them, a final 0 appended."""

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np
        self.num_inference_steps = num_inference_steps
        T = self.config["num_train_timesteps"]
        ts = np.linspace(0, T - 1, num_inference_steps, dtype=float)[::-1].copy()
        acp = self.alphas_cumprod.numpy().astype(np.float64)
        sig = ((1 - acp) / acp) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32)).to(device)
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.float32 if device is not None and torch.device(device).type == "mps" else torch.float64)
        self.init_noise_sigma = float(self.sigmas.max())
        self._reset()

    def _reset(self):
        pass

    def _index(self, timestep):
        t = float(timestep)
        return int((self.timesteps.double() - t).abs().argmin())

    def scale_model_input(self, sample, timestep=None):
        s = float(self.sigmas[self._index(timestep)])
        return sample / ((s * s + 1) ** 0.5)

    def _x0(self, model_output, sample, s):
        pt = self.config["prediction_type"]
        if pt == "epsilon":
            return sample - s * model_output
        if pt == "v_prediction":
            return model_output * (-s / (s * s + 1) ** 0.5) + sample / (s * s + 1)
        raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, or `v_prediction`")


class EulerDiscreteScheduler(_SigmaBase):
    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        i = self._index(timestep)
        s, s_next = float(self.sigmas[i]), float(self.sigmas[i + 1])
        x0 = self._x0(model_output, sample, s)
        prev = sample + (sample - x0) / s * (s_next - s)                 # s_churn = 0: sigma_hat = sigma
        return SchedulerOutput(prev, x0) if return_dict else (prev,)


class EulerAncestralDiscreteScheduler(_SigmaBase):
    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        i = self._index(timestep)
        s, s_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        x0 = self._x0(model_output, sample, s)
        s_up = (s_to ** 2 * (s ** 2 - s_to ** 2) / s ** 2) ** 0.5
        s_down = (s_to ** 2 - s_up ** 2) ** 0.5
        prev = sample + (sample - x0) / s * (s_down - s)
        gdev = generator.device if generator is not None else model_output.device
        noise = torch.randn(model_output.shape, dtype=model_output.dtype, device=gdev, generator=generator).to(model_output.device)
        prev = prev + noise * s_up
        return SchedulerOutput(prev, x0) if return_dict else (prev,)


class LMSDiscreteScheduler(_SigmaBase):
    def _reset(self):
        self.derivatives = []

    def get_lms_coefficient(self, order, t, current_order):
        from scipy import integrate
        sig = self.sigmas.tolist()

        def lms_derivative(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - sig[t - k]) / (sig[t - current_order] - sig[t - k])
            return prod
        return integrate.quad(lms_derivative, sig[t], sig[t + 1], epsrel=1e-4)[0]

    def step(self, model_output, timestep, sample, order: int = 4, return_dict=True):
        i = self._index(timestep)
        s = float(self.sigmas[i])
        x0 = self._x0(model_output, sample, s)
        self.derivatives.append((sample - x0) / s)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self.get_lms_coefficient(order, i, c) for c in range(order)]
        prev = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        return SchedulerOutput(prev, x0) if return_dict else (prev,)


class PNDMScheduler(_Base):
    """PLMS as Stable Diffusion configures it (skip_prk_steps=True): linear multistep on the eps history; the second
    timestep is visited twice (n + 1 model calls for n steps)."""

    def __init__(self, *a, skip_prk_steps=False, set_alpha_to_one=False, steps_offset=0, **kw):
        super().__init__(*a, skip_prk_steps=skip_prk_steps, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **kw)
        if not skip_prk_steps:
            raise NotImplementedError("the Runge-Kutta warm-up (skip_prk_steps=False) is not built; Stable Diffusion configs skip it")
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self._acp = [float(a) for a in self.alphas_cumprod]

    @classmethod
    def stable_diffusion(cls, prediction_type="epsilon", **kw):
        return super().stable_diffusion(prediction_type, skip_prk_steps=True, steps_offset=1, **kw)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config["num_train_timesteps"] // num_inference_steps
        ts = [i * ratio + self.config["steps_offset"] for i in range(num_inference_steps)]
        plms = (ts[:-1] + ts[-2:-1] + ts[-1:])[::-1]
        self.timesteps = torch.tensor(plms, dtype=torch.int64, device=device)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def step(self, model_output, timestep, sample, return_dict=True):
        t = int(timestep)
        ratio = self.config["num_train_timesteps"] // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t, t = t, t + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = (model_output + e[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            model_output = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            model_output = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        a_t = self._acp[t]
        a_prev = self._acp[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.config["prediction_type"] == "v_prediction":
            model_output = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        prev = (a_prev / a_t) ** 0.5 * sample - (a_prev - a_t) * model_output / denom
        self.counter += 1
        return SchedulerOutput(prev) if return_dict else (prev,)


class DPMSolverMultistepScheduler(_Base):
    """DPM-Solver++ (2M, midpoint, lower_order_final) — the configuration inference.py's "dpm_solver++" gets from the SD
    scheduler config with diffusers 0.14 defaults (solver_order=2, algorithm_type="dpmsolver++", no thresholding)."""

    def __init__(self, *a, solver_order=2, lower_order_final=True, **kw):
        super().__init__(*a, solver_order=solver_order, lower_order_final=lower_order_final, **kw)
        if solver_order not in (1, 2):
            raise NotImplementedError("solver_order 3 is not built")
        acp = self.alphas_cumprod.double()
        self.alpha_t, self.sigma_t = acp.sqrt(), (1 - acp).sqrt()
        self.lambda_t = self.alpha_t.log() - self.sigma_t.log()

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np
        self.num_inference_steps = num_inference_steps
        T = self.config["num_train_timesteps"]
        ts = np.linspace(0, T - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.model_outputs, self.lower_order_nums, self._prev_ts = [], 0, []

    def _x0(self, model_output, t, sample):
        a, s = float(self.alpha_t[t]), float(self.sigma_t[t])
        pt = self.config["prediction_type"]
        if pt == "epsilon":
            return (sample - s * model_output) / a
        if pt == "v_prediction":
            return a * sample - s * model_output
        return model_output

    def step(self, model_output, timestep, sample, return_dict=True):
        import math
        t = int(timestep)
        ts = self.timesteps.tolist()
        i = ts.index(t)
        prev_t = 0 if i == len(ts) - 1 else ts[i + 1]
        final = i == len(ts) - 1 and self.config["lower_order_final"] and len(ts) < 15
        x0 = self._x0(model_output, t, sample)
        self.model_outputs = (self.model_outputs + [x0])[-2:]
        self._prev_ts = (self._prev_ts + [t])[-2:]
        lam, al, sg = self.lambda_t, self.alpha_t, self.sigma_t
        h = float(lam[prev_t] - lam[t])
        c = float(al[prev_t]) * (math.exp(-h) - 1.0)
        prev = float(sg[prev_t] / sg[t]) * sample - c * x0
        if not (self.config["solver_order"] == 1 or self.lower_order_nums < 1 or final):
            s1 = self._prev_ts[-2]
            r0 = float(lam[t] - lam[s1]) / h
            prev = prev - 0.5 * c * (1.0 / r0) * (self.model_outputs[-1] - self.model_outputs[-2])
        if self.lower_order_nums < self.config["solver_order"]:
            self.lower_order_nums += 1
        return SchedulerOutput(prev, x0) if return_dict else (prev,)


SCHEDULER_MAPPING = {               # inference.py:60-67
    "ddim": DDIMScheduler,
    "plms": PNDMScheduler,
    "lms": LMSDiscreteScheduler,
    "euler": EulerDiscreteScheduler,
    "euler_ancestral": EulerAncestralDiscreteScheduler,
    "dpm_solver++": DPMSolverMultistepScheduler,
}
