"""CPU: the samplers of e4t/schedulers.py (diffusers is not installed, so they cannot be pinned against it) are checked on
a problem with a closed-form answer.  Data ~ N(0, s^2 I): the optimal noise prediction at signal level abar is
eps*(x, abar) = sqrt(1-abar) x / (abar s^2 + 1 - abar), and the probability-flow ODE every deterministic sampler integrates
maps x(abar_T) to x(abar) = x(abar_T) * sqrt(abar s^2 + 1 - abar) / sqrt(abar_T s^2 + 1 - abar_T).  Each sampler, driven
exactly like the pipeline drives it (init_noise_sigma, scale_model_input, step(...).prev_sample over .timesteps), must land
on that value within its order of accuracy (measured: DDIM / Euler first order, 1e-2 at 200 steps, halving per doubling;
PLMS 1e-4 at 100 steps for eps — first order for v, where diffusers extrapolates the raw v history; DPM-Solver++(2M) 4e-3
at 200; LMS 1e-4 at 50) and get worse with fewer steps; the ancestral sampler must reproduce the data variance."""
import math

import pytest
import torch

from test_unet_host_logic import emu_fp32  # noqa: F401  (DDIM's linear update goes through the op backend)

S_DATA = 0.6


def drive(sch, n, x_T_noise, v_pred=False):
    sch.set_timesteps(n)
    x = x_T_noise * sch.init_noise_sigma
    acp = sch.alphas_cumprod.double()
    for i, t in enumerate(sch.timesteps):
        if hasattr(sch, "sigmas"):
            sg = float(sch.sigmas[i])
            a = 1.0 / (1.0 + sg * sg)
        else:
            a = float(acp[int(t)])
        inp = sch.scale_model_input(x, t)
        eps = math.sqrt(1 - a) * inp / (a * S_DATA ** 2 + 1 - a)
        if v_pred:                                   # v = sqrt(a) eps - sqrt(1-a) x0, with x0 = (x - sqrt(1-a) eps) / sqrt(a)
            x0 = (inp - math.sqrt(1 - a) * eps) / math.sqrt(a)
            out = math.sqrt(a) * eps - math.sqrt(1 - a) * x0
        else:
            out = eps
        x = sch.step(out, t, x).prev_sample
    return x


def exact(sch, n, x_T_noise):
    """closed-form end point for the same start"""
    sch.set_timesteps(n)
    acp = sch.alphas_cumprod.double()
    if hasattr(sch, "sigmas"):
        sg = float(sch.sigmas[0])
        a_T, a_end = 1.0 / (1.0 + sg * sg), 1.0
        x_vp_T = x_T_noise * sg * math.sqrt(a_T)
    else:
        a_T = float(acp[int(sch.timesteps[0])])
        a_end = float(acp[0])                        # set_alpha_to_one=False / DPM-Solver's final timestep 0
        x_vp_T = x_T_noise
    f = lambda a: math.sqrt(a * S_DATA ** 2 + 1 - a)
    return x_vp_T * f(a_end) / f(a_T)


@pytest.mark.parametrize("name,steps,tol", [("ddim", 200, 1.5e-2), ("plms", 100, 2e-2), ("dpm_solver++", 200, 6e-3), ("euler", 200, 1.8e-2), ("lms", 50, 5e-4)])
@pytest.mark.parametrize("v_pred", [False, True])
def test_deterministic_samplers_reach_the_ode_solution(emu_fp32, name, steps, tol, v_pred):
    from e4t.schedulers import SCHEDULER_MAPPING
    sch = SCHEDULER_MAPPING[name].stable_diffusion("v_prediction" if v_pred else "epsilon")
    noise = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    got = drive(sch, steps, noise, v_pred)
    want = exact(sch, steps, noise)
    err = float((got - want).norm() / want.norm())
    assert err < tol, (name, err)
    # fewer steps -> larger error (the update really integrates, it is not an identity that happens to fit)
    coarse = float((drive(sch, max(steps // 10, 4), noise, v_pred) - exact(sch, max(steps // 10, 4), noise)).norm() / want.norm())
    assert coarse > err


def test_ancestral_sampler_reproduces_the_data_variance():
    from e4t.schedulers import EulerAncestralDiscreteScheduler
    ratios = []
    for n in (30, 200):
        sch = EulerAncestralDiscreteScheduler.stable_diffusion()
        noise = torch.randn(64, 4, 16, 16, generator=torch.Generator().manual_seed(1))
        torch.manual_seed(2)
        out = drive(sch, n, noise)
        assert abs(float(out.mean())) < 0.01
        ratios.append(float(out.std()) / S_DATA)
    assert abs(ratios[1] - 1.0) < 0.04 and abs(ratios[1] - 1.0) < abs(ratios[0] - 1.0)      # measured 0.878 -> 0.974 (first-order bias)


def test_scheduler_surface():
    from e4t.schedulers import SCHEDULER_MAPPING, PNDMScheduler
    assert sorted(SCHEDULER_MAPPING) == ["ddim", "dpm_solver++", "euler", "euler_ancestral", "lms", "plms"]       # inference.py:60-67
    p = PNDMScheduler.stable_diffusion()
    p.set_timesteps(50)
    assert p.timesteps[:4].tolist() == [981, 961, 961, 941] and len(p.timesteps) == 51
    with pytest.raises(NotImplementedError):
        PNDMScheduler()                              # diffusers' default (Runge-Kutta warm-up) is not built
    for name, cls in SCHEDULER_MAPPING.items():
        s = cls.from_config(dict(cls.stable_diffusion().config, _class_name="X"))
        s.set_timesteps(7)
        assert len(s) == 1000 and s.order == 1 and len(s.timesteps) in (7, 8)
