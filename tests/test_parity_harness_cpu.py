"""CPU: the whole-step parity harness (tests/parity_step.py) itself, with the bf16-rounding op emulation standing in for the HIP
kernels as the "native" leg.  Same builders, same key maps, same kink alignment, same rule as the GPU tests — so a change to the
trainer, the modules' host logic or the harness that would break `-m gpu` parity shows up here first, in seconds.  (This is NOT a
parity claim for the kernels: those are compared on the GPU, tests/test_fullsize_gpu.py / test_configs_gpu.py / smoke.)"""
import pytest
import torch


@pytest.fixture()
def emu_bf16():
    from e4t import ops
    from emu_backend import EmuBackend
    old_b, old_act = ops.set_backend(EmuBackend(round_bf16=True)), ops.ACT
    ops.ACT = torch.bfloat16
    yield ops
    ops.set_backend(old_b)
    ops.ACT = old_act


@pytest.mark.parametrize("case", ["tiny_sd1", "tiny_sd2", "tuning_tiny", "unfrozen_vit_tiny"])
def test_emulated_product_path_passes_the_calibrated_rule(emu_bf16, case):
    import parity_step as ps
    rep = ps.run(case, torch.device("cpu"), verbose=False)
    assert rep["n_bad"] == 0, rep["bad"][:5]
    assert rep["losses"]["count"] == 2 and rep["enc_maps"]["count"] == 13
    assert rep["grads"]["count"] > (400 if case == "tuning_tiny" else 150)
    if case.startswith("unfrozen"):
        assert rep["grads"]["by_part"][".clip_vision."] >= 12
