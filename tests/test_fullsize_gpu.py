"""-m gpu: BASELINE.json configs[1] at B=1 — the FULL SD-1.4 UNet + ViT-H-14 E4T encoder + CLIP-L text encoder + AutoencoderKL
encoder at 512 px — one training step on the HIP kernels vs the CPU fp32 oracle, with the autocast-calibrated tolerance of
SURVEY.md §8(c) (tests/parity_step.py).  This is the reference's own full-size smoke (pretrain_e4t.py:595-654,
e4t/encoder.py:171-296) turned into a parity test: VAE latents, the 13 encoder maps, the domain embedding, both losses and the
gradient of every one of the 96 x 9 weight-offset tensors and of every E4T-head parameter."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_sd14_step_matches_oracle(hip_env):
    import parity_step
    rep = parity_step.run("full_sd14", torch.device("cuda:0"))
    assert rep["n_bad"] == 0
    assert rep["grads"]["count"] >= 96 * 2 + 5          # every weight-offset instance's two big matrices + the head
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_full_sd14.json"), "w") as fh:
            json.dump(rep, fh, indent=1)


def test_full_sd14_batch16_step_matches_sixteen_single_sample_steps(hip_env):
    """BASELINE configs[1] at the batch bench.py times (B = 16): the kernel instantiations launch_gemm() picks at M = 16 x tokens
    (256-row tiles, other split-K factors, other attention grids) against the oracle-pinned B = 1 path, sample by sample."""
    import parity_step
    rep = parity_step.batch_consistency("full_sd14", torch.device("cuda:0"), B=16)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_full_sd14_batch16.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    assert rep["n_bad"] == 0, rep["bad"]


def test_tuning_full_batch16_step_matches_sixteen_single_sample_steps(hip_env):
    """BASELINE configs[3] at the size bench.py's `secondary` block times it (tuning_e4t.py:266-338): full SD-1.4 UNet on 64 x 64 latents,
    every UNet weight trainable — 3x3-conv dW through im2col + split-K TN GEMM at M = 16 x 4096 rows — ViT-H-14, B = 16, against sixteen
    B = 1 steps of the same samples (the ORACLE leg of the tuning step is `tuning_real_width`, test_configs_gpu.py).  Was an offline
    tool call in round 4 (tools/gpu_r04_a.sh); 27 s."""
    import parity_step
    rep = parity_step.batch_consistency("tuning_full", torch.device("cuda:0"), B=16)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_tuning_full_batch16.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    assert rep["n_bad"] == 0, rep["bad"]
    assert rep["n_quantities"] > 1700           # every UNet parameter's gradient, not only the weight offsets
