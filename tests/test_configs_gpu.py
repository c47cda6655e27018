"""-m gpu: model-level parity of the configurations BASELINE.json names besides the headline one, on the real kernels vs
the CPU fp32 oracle with the autocast-calibrated tolerance (tests/parity_step.py):

  configs[3]  tuning_e4t.py:139-147,266-338 — every UNet weight trains: linear dW, 3x3-conv dW (im2col + TN GEMM), GN / LN
              affine gradients, dW = dW_eff o (1 + offsets); global gradient norm (the clip's input)
  configs[4]  the SD-2.x UNet config — heads 5/10/20/20 (dh = 64), ctx 1024, use_linear_projection
              (cross_attention.py:224-226, transformer_2d.py:151,258-261), v-prediction (pretrain_e4t.py:640-641)
  README      --unfreeze_clip_vision (encoder.py:98-99): the backward through the ViT tower

Each at the real channel widths (small spatial size so the CPU oracle takes seconds) and at a tiny width."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(name):
    import parity_step
    rep = parity_step.run(name, torch.device("cuda:0"))
    assert rep["n_bad"] == 0
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"parity_{name}.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    return rep


@pytest.mark.parametrize("name", ["tuning_tiny", "tuning_real_width"])
def test_tuning_step_matches_oracle(hip_env, name):
    rep = _run(name)
    assert rep["grads"]["count"] > 500          # every UNet parameter was compared, not only the weight offsets
    assert rep["other"]["count"] >= 2           # domain embedding + global gradient norm


@pytest.mark.parametrize("name", ["tiny_sd2", "sd2_real_width", "full_sd21"])
def test_sd2_config_step_matches_oracle(hip_env, name):
    """full_sd21: BASELINE configs[4] at its real size — SD-2.x UNet config (ctx 1024, dh 64, linear projections, v-prediction) on
    96 x 96 latents (T = 9216 self-attention), ViT-H-14, 23-layer text encoder, AutoencoderKL encoder at 768 px, B = 1."""
    _run(name)


@pytest.mark.parametrize("name", ["unfrozen_vit_tiny", "unfrozen_vit", "unfrozen_vit_full"])
def test_unfrozen_vit_step_matches_oracle(hip_env, name):
    """unfrozen_vit_full: the README's --unfreeze_clip_vision recipe with the whole 32-layer ViT-H-14 tower trainable"""
    rep = _run(name)
    assert rep["grads"]["by_part"][".clip_vision."] >= (32 * 12 if name == "unfrozen_vit_full" else 12)    # the tower's gradients were compared, not only the head's
