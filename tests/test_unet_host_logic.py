"""CPU: the native UNet's host-side graph (module wiring, NHWC plumbing, fused concat, weight-offset
banks, hand-written backward of every op) driven through the fp32 op emulation and compared against the
fp32 oracle with one shared state dict.  Validates everything except the HIP kernels themselves (those are
checked op-by-op on the GPU in test_kernels_gpu.py)."""
import pytest
import torch

import e4t_oracle as orc
from emu_backend import EmuBackend


@pytest.fixture()
def emu_fp32():
    from e4t import ops
    old_b, old_act = ops.set_backend(EmuBackend(round_bf16=False)), ops.ACT
    ops.ACT = torch.float32
    yield ops
    ops.set_backend(old_b)
    ops.ACT = old_act


def build_pair(cfg, seed=0):
    from e4t.models.unet_2d_condition import UNet2DConditionModel
    torch.manual_seed(seed)
    ref = orc.UNet2DConditionModel(**cfg)
    nat = UNet2DConditionModel(**cfg)
    sd = ref.state_dict()
    assert set(sd) == set(nat.state_dict()), set(sd) ^ set(nat.state_dict())
    nat.load_state_dict(sd)
    return ref, nat


@pytest.mark.parametrize("linproj", [False, True])
def test_unet_forward_backward_matches_oracle(emu_fp32, linproj):
    cfg = dict(orc.tiny_unet_config(ctx_dim=64), use_linear_projection=linproj)
    ref, nat = build_pair(cfg)
    for m in (ref, nat):   # pretrain: only weight-offset parameters train
        for n, p in m.named_parameters():
            p.requires_grad_("wo" in n)
    g = torch.Generator().manual_seed(1)
    B = 2
    x = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([3, 977])
    ctx_r = torch.randn(B, 7, 64, generator=g, requires_grad=True)
    ctx_n = ctx_r.detach().clone().requires_grad_(True)

    enc_r = ref(x, t, ctx_r.detach(), return_encoder_outputs=True)["down_block_samples"]
    enc_n = nat(x, t, ctx_n.detach(), return_encoder_outputs=True)["down_block_samples"]
    assert len(enc_n) == 13
    for a, b in zip(enc_n, enc_r):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)

    # two passes sharing the down/mid weight offsets, as in pretrain_e4t.py:622-636
    w = torch.randn(B, 4, 16, 16, generator=g)
    pooled_r = torch.cat([s.mean(dim=(2, 3)) for s in ref(x, t, ctx_r, return_encoder_outputs=True)["down_block_samples"]], -1)
    loss_r = (ref(x, t, ctx_r) * w).sum() + pooled_r.pow(2).sum()
    loss_r.backward()
    pooled_n = torch.cat([s.mean(dim=(2, 3)) for s in nat(x, t, ctx_n, return_encoder_outputs=True)["down_block_samples"]], -1)
    out_n = nat(x, t, ctx_n).sample
    loss_n = (out_n * w).sum() + pooled_n.pow(2).sum()
    torch.testing.assert_close(loss_n, loss_r, rtol=2e-4, atol=2e-4)
    loss_n.backward()
    torch.testing.assert_close(ctx_n.grad, ctx_r.grad, rtol=2e-3, atol=2e-4)
    gr = dict(ref.named_parameters())
    checked = 0
    for n, p in nat.named_parameters():
        if "wo" in n:
            assert p.grad is not None, n
            torch.testing.assert_close(p.grad, gr[n].grad, rtol=5e-3, atol=2e-4, msg=lambda m, n=n: f"{n}: {m}")
            checked += 1
    assert checked == 16 * 2 * 3 * 9


def test_second_step_reuses_and_refreshes_offsets(emu_fp32):
    """W_eff is cached across the two passes of a step and refreshed when the parameters change."""
    cfg = orc.tiny_unet_config(ctx_dim=64)
    ref, nat = build_pair(cfg, seed=3)
    x = torch.randn(1, 4, 16, 16)
    t = torch.tensor([10])
    ctx = torch.randn(1, 5, 64)
    y0 = nat(x, t, ctx).sample
    with torch.no_grad():
        for n, p in list(nat.named_parameters()):
            if n.endswith("wo_q.v"):
                p.add_(0.5)
        for n, p in list(ref.named_parameters()):
            if n.endswith("wo_q.v"):
                p.add_(0.5)
    y1 = nat(x, t, ctx).sample
    assert (y1 - y0).abs().max() > 1e-4
    torch.testing.assert_close(y1, ref(x, t, ctx), rtol=2e-4, atol=2e-4)


def test_standalone_weightoffsets_module(emu_fp32):
    from e4t.weightoffsets import WeightOffsets
    torch.manual_seed(5)
    ref = orc.WeightOffsets(48, 80)
    nat = WeightOffsets(48, 80)
    nat.load_state_dict(ref.state_dict())
    g = torch.randn(80, 48)
    o_r = ref(); o_r.backward(g)
    o_n = nat(); o_n.backward(g)
    torch.testing.assert_close(o_n, o_r, rtol=1e-5, atol=1e-6)
    for (n, a), (_, b) in zip(nat.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-5, msg=lambda m, n=n: f"{n}: {m}")


def test_tuning_mode_all_unet_gradients(emu_fp32):
    """tuning_e4t.py:139-147: every UNet parameter trains — conv weight gradients (gather-transpose + GEMM), GN/LN affine
    gradients, plain linear weight/bias gradients, and dW = dW_eff o (1 + offsets) for the modulated projections."""
    cfg = orc.tiny_unet_config(ctx_dim=64)
    ref, nat = build_pair(cfg, seed=4)
    g = torch.Generator().manual_seed(9)
    B = 2
    x = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([500, 20])
    ctx = torch.randn(B, 6, 64, generator=g)
    w = torch.randn(B, 4, 16, 16, generator=g)
    (ref(x, t, ctx) * w).sum().backward()
    (nat(x, t, ctx).sample * w).sum().backward()
    gr = dict(ref.named_parameters())
    n = 0
    for name, p in nat.named_parameters():
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad, gr[name].grad, rtol=5e-3, atol=3e-4, msg=lambda m, name=name: f"{name}: {m}")
        n += 1
    assert n == len(gr)
