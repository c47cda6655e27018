"""CPU: one full pre-training step of the native trainer (flat params, in-place grads, fused AdamW, two UNet
passes, embed injection through the text encoder) vs the oracle step + torch.optim.AdamW — op emulation in fp32."""
import copy

import pytest
import torch

import e4t_oracle as orc
from test_unet_host_logic import emu_fp32  # noqa: F401
from test_encoder_host_logic import TINY_VIT, BOC

TEXT_CFG = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=128, max_len=9, act="quick_gelu")


def build(seed=0):
    from e4t.encoder import E4TEncoder
    from torch_twins import CLIPTextModel
    from e4t.models.unet_2d_condition import UNet2DConditionModel
    torch.manual_seed(seed)
    cfg = orc.tiny_unet_config(ctx_dim=64)
    r_unet = orc.UNet2DConditionModel(**cfg)
    r_enc = orc.E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, vit_cfg=TINY_VIT)
    text = CLIPTextModel(**TEXT_CFG).requires_grad_(False)
    n_unet = UNet2DConditionModel(**cfg)
    n_unet.load_state_dict(r_unet.state_dict())
    n_enc = E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3)
    n_enc.load_state_dict(r_enc.state_dict())
    return r_unet, r_enc, n_unet, n_enc, text


def test_one_step_matches_oracle(emu_fp32):
    from e4t.trainer import E4TTrainer
    r_unet, r_enc, n_unet, n_enc, text = build()
    B = 2
    g = torch.Generator().manual_seed(7)
    pixels = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    latents = torch.randn(B, 4, 16, 16, generator=g) * 0.18215
    noise = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([5, 700])
    ids = torch.randint(1, 99, (B, 9), generator=g)
    pidx = torch.tensor([2, 4])
    lr = 1e-3

    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=lr, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long),
                    device=torch.device("cpu"))
    # oracle side
    for n, p in r_unet.named_parameters():
        p.requires_grad_("wo" in n)
    params = orc.trainable_parameters(r_unet, r_enc)
    opt = torch.optim.AdamW(params, lr=lr)
    acp = orc.ddpm_alphas_cumprod()
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([11]))[0]
        ctx0 = text(input_ids=torch.zeros(1, 9, dtype=torch.long))[0]
        emb = text.get_input_embeddings()(ids)
    loss_r, ld_r, lr_r, _ = orc.e4t_losses(r_unet, r_enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds)[0], pixels, latents, noise, t,
                                          emb, pidx.tolist(), ctx0, class_embed, acp)
    loss_r.backward()
    before = {n: p.detach().clone() for n, p in r_unet.named_parameters() if "wo" in n}
    opt.step()

    loss_n, ld_n, lr_n = tr.train_step(pixels, ids, pidx, noise=noise, timesteps=t, latents=latents)
    torch.testing.assert_close(ld_n, ld_r.detach(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(lr_n, lr_r.detach(), rtol=1e-3, atol=1e-5)
    # parameters after AdamW.  Adam's first step moves every coordinate by ~lr*sign(g): compare the UPDATE direction
    # where |g| is not tiny, and the values everywhere.
    nat = dict(n_unet.named_parameters())
    moved = 0
    for n, p in r_unet.named_parameters():
        if "wo" not in n:
            continue
        torch.testing.assert_close(nat[n].data, p.data, rtol=0, atol=2.5 * lr, msg=lambda m, n=n: f"{n}: {m}")
        moved += int((p.data - before[n]).abs().max() > 0)
    assert moved == 16 * 2 * 3 * 9
    ne = dict(n_enc.named_parameters())
    for n, p in r_enc.named_parameters():
        if p.requires_grad:
            big = p.grad.abs() > 1e-6
            torch.testing.assert_close(ne[n].data[big], p.data[big], rtol=0, atol=1e-4, msg=lambda m, n=n: f"{n}: {m}")
    # gradients were cleared in place and storage is still the flat buffer
    assert float(tr.flat.grad.abs().max()) == 0.0
    assert n_enc.first_linears[2].weight.grad.data_ptr() >= tr.flat.grad.data_ptr()
    # a second step runs (weight-offset caches refresh through the weights epoch)
    loss2, _, _ = tr.train_step(pixels, ids, pidx, noise=noise, timesteps=t, latents=latents)
    assert torch.isfinite(loss2)


def test_shared_prefix_is_result_preserving(emu_fp32):
    """SURVEY §8a restructuring (3): computing the context-independent UNet prefix once for the two passes of a step
    must leave the losses and every trainable gradient unchanged."""
    from e4t.trainer import E4TTrainer
    B = 2
    g = torch.Generator().manual_seed(3)
    pixels = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    latents = torch.randn(B, 4, 16, 16, generator=g) * 0.18215
    noise = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([17, 912])
    ids = torch.randint(1, 99, (B, 9), generator=g)
    pidx = torch.tensor([1, 5])
    res = []
    for share in (False, True):
        _, _, n_unet, n_enc, text = build()
        tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long),
                        device=torch.device("cpu"))
        tr.share_prefix = share
        calls = []
        blk = n_unet.down_blocks[0].attentions[0].transformer_blocks[0]
        orig = blk.attn1.forward
        blk.attn1.forward = lambda *a, _o=orig, **k: (calls.append(1), _o(*a, **k))[1]
        loss, ld, lr_ = tr.losses(pixels, latents, noise, t, ids, pidx)
        loss.backward()
        assert len(calls) == (1 if share else 2)          # the first self-attention really ran once / twice
        assert n_unet._prefix.store == {} and not n_unet.share_prefix
        res.append((ld.detach().clone(), lr_.detach().clone(), tr.flat.grad.detach().clone()))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-4, atol=1e-7)


def test_inplace_param_grads_match_autograd(emu_fp32):
    """Tuning (every UNet weight trainable): weight / bias gradients accumulated by the producing kernels straight into the
    trainer's flat buffer (functional.set_inplace_param_grads) must equal what autograd's AccumulateGrad produces."""
    from e4t import functional as Fn
    from e4t.trainer import E4TTrainer
    B = 2
    g = torch.Generator().manual_seed(5)
    pixels = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    latents = torch.randn(B, 4, 16, 16, generator=g) * 0.18215
    noise = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([301, 44])
    ids = torch.randint(1, 99, (B, 9), generator=g)
    pidx = torch.tensor([3, 6])
    res = []
    for inplace in (False, True):
        _, _, n_unet, n_enc, text = build()
        tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=1e-3, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long),
                        device=torch.device("cpu"), tuning=True)
        loss, _, _ = tr.losses(pixels, latents, noise, t, ids, pidx)
        Fn.set_inplace_param_grads(inplace)
        try:
            loss.backward()
        finally:
            Fn.set_inplace_param_grads(False)
        res.append(tr.flat.grad.detach().clone())
        assert float(res[-1].abs().max()) > 0
    torch.testing.assert_close(res[0], res[1], rtol=1e-4, atol=1e-7)


def _data(B, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(px=torch.rand(B, 3, 64, 64, generator=g) * 2 - 1, lat=torch.randn(B, 4, 16, 16, generator=g) * 0.18215,
                noise=torch.randn(B, 4, 16, 16, generator=g), t=torch.randint(0, 1000, (B,), generator=g), ids=torch.randint(1, 99, (B, 9), generator=g),
                pidx=torch.randint(1, 8, (B,), generator=g))


def _trainer(lr=1e-3):
    from e4t.trainer import E4TTrainer
    r_unet, r_enc, n_unet, n_enc, text = build()
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=lr, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long), device=torch.device("cpu"))
    return tr, (r_unet, r_enc, text)


def _step(tr, d, **kw):
    return tr.train_step(d["px"], d["ids"], d["pidx"], noise=d["noise"], timesteps=d["t"], latents=d["lat"], **kw)


def test_gradient_accumulation_matches_oracle(emu_fp32):
    """two micro-batches with loss/2 each, one optimiser step (accelerator.accumulate + accelerator.backward semantics)"""
    tr, (r_unet, r_enc, text) = _trainer()
    for n, p in r_unet.named_parameters():
        p.requires_grad_("wo" in n)
    params = orc.trainable_parameters(r_unet, r_enc)
    opt = torch.optim.AdamW(params, lr=1e-3)
    acp = orc.ddpm_alphas_cumprod()
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([11]))[0]
        ctx0 = text(input_ids=torch.zeros(1, 9, dtype=torch.long))[0]
    micro = [_data(1, 21), _data(1, 22)]
    for d in micro:
        with torch.no_grad():
            emb = text.get_input_embeddings()(d["ids"])
        loss, _, _, _ = orc.e4t_losses(r_unet, r_enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds)[0], d["px"], d["lat"], d["noise"], d["t"],
                                       emb, d["pidx"].tolist(), ctx0, class_embed, acp)
        (loss / 2).backward()
    opt.step()
    before = tr.flat.data.clone()
    _step(tr, micro[0], sync=False, loss_scale=0.5)
    assert torch.equal(tr.flat.data, before) and tr.step_count == 0 and float(tr.flat.grad.abs().sum()) > 0      # nothing applied yet
    _step(tr, micro[1], sync=True, loss_scale=0.5)
    assert tr.step_count == 1 and float(tr.flat.grad.abs().sum()) == 0
    want = dict(r_unet.named_parameters())
    checked = 0
    for n, p in tr.unet.named_parameters():
        if "wo" in n and n.endswith("linear_row.weight"):
            d = (p.detach() - want[n].detach()).abs()
            assert float((d > 5e-6).float().mean()) < 2e-3, n           # Adam's first step: sign flips where |g| is rounding noise
            checked += 1
    assert checked == 96


def test_adamw_in_two_parts_around_a_deferred_region_changes_no_bit(emu_fp32):
    """Data parallel: AdamW of everything outside region D runs under D's all-reduce, D follows (optimizer_step(deferred=...)); the update
    is element-wise, so the split must give the parameters and moments of the one-launch update bit for bit."""
    class Handle:
        waited = 0

        def wait(self):
            Handle.waited += 1
    res = []
    for split in (False, True):
        tr, _ = _trainer()
        loss, _, _ = tr.losses(*[_data(2, 31)[k] for k in ("px", "lat", "noise", "t", "ids", "pidx")])
        loss.backward()
        n = tr.flat.numel
        lo, hi = (n // 3) // 64 * 64, (2 * n // 3) // 64 * 64
        tr.optimizer_step(((lo, hi), [Handle(), Handle()]) if split else None)
        res.append((tr.flat.data.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.step_count))
    assert Handle.waited == 2 and res[0][3] == res[1][3] == 1
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    assert float((res[0][1] != 0).float().mean()) > 0.5         # (the gradients were there)


def test_head_stack_update_from_factors_equals_the_materialised_gradient(emu_fp32):
    """Round 6: a synchronising step hands the two factors of the head's stacked weight gradient (dW_i = gb^T z_i) to AdamW
    (e4t_adamw_rank) and never writes the stack.  Same parameters / moments as the materialised path (a different summation order
    only), the stack region of the flat gradient stays zero and is not cleared, and whatever needs the stack declines the factors:
    accumulated micro-batches, a gradient clip."""
    batches = [_data(2, 41), _data(2, 42)]
    res, took = [], []
    for factored in (True, False):
        tr, _ = _trainer()
        tr.factored_head_update = factored
        n, rows, cols = tr._stack_shape
        w_end = n * rows * cols
        seen = []
        orig = tr._take_head_factors
        tr.encoder.take_head_factors = lambda gb, Z, orig=orig, seen=seen: (seen.append(orig(gb, Z)), seen[-1])[1]
        for d in batches:
            _step(tr, d)
            assert float(tr.flat.grad.abs().sum()) == 0 and tr.encoder._stack_grad_is_zero and tr._head_factors is None
        took.append(seen)
        res.append((tr.flat.data.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), w_end))
    assert took == [[True, True], [False, False]]
    w_end = res[0][3]
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a[w_end:], b[w_end:])                    # everything behind the stack: the same launches
        torch.testing.assert_close(a[:w_end], b[:w_end], rtol=2e-4, atol=2e-6)
    assert float((res[0][1][:w_end] != 0).float().mean()) > 0.5     # (the stack did get gradients)

    # a micro-batch that only accumulates writes the stack; the synchronising step behind it must not take factors
    tr, _ = _trainer()
    ref, _ = _trainer()
    ref.factored_head_update = False
    for t in (tr, ref):
        _step(t, batches[0], sync=False, loss_scale=0.5)
        assert t._accum_pending and not t.encoder._stack_grad_is_zero
        _step(t, batches[1], sync=True, loss_scale=0.5)
        assert not t._accum_pending and t.encoder._stack_grad_is_zero and float(t.flat.grad.abs().sum()) == 0
    assert torch.equal(tr.flat.data, ref.flat.data) and torch.equal(tr.exp_avg, ref.exp_avg)

    # with a gradient clip the norm is taken over the materialised gradient
    tr, _ = _trainer()
    tr.max_grad_norm = 1.0
    assert tr._take_head_factors(None, None) is False


def test_training_state_round_trip_resumes_bitwise(emu_fp32, tmp_path):
    batches = [_data(2, 31 + i) for i in range(3)]
    tr, _ = _trainer()
    for d in batches:
        _step(tr, d)
    full = tr.flat.data.clone()
    tr2, _ = _trainer()
    for d in batches[:2]:
        _step(tr2, d)
    torch.save(tr2.state_dict(), tmp_path / "state.pt")
    tr3, _ = _trainer()
    tr3.load_state_dict(torch.load(tmp_path / "state.pt"))
    assert tr3.step_count == 2
    _step(tr3, batches[2])
    assert torch.equal(tr3.flat.data, full)
    tr4, _ = _trainer()
    sd = tr2.state_dict()
    sd["params"] = sd["params"][:-1]
    with pytest.raises(RuntimeError, match="parameters"):
        tr4.load_state_dict(sd)


def test_lr_schedules():
    import math
    from e4t.optimization import LRSchedule, get_lr_lambda
    assert [get_lr_lambda("constant")(s) for s in (0, 10)] == [1.0, 1.0]
    f = get_lr_lambda("constant_with_warmup", 4)
    assert [f(s) for s in (0, 2, 4, 9)] == [0.0, 0.5, 1.0, 1.0]
    f = get_lr_lambda("linear", 10, 110)
    assert f(5) == 0.5 and f(10) == 1.0 and f(60) == 0.5 and f(110) == 0.0 and f(200) == 0.0
    f = get_lr_lambda("cosine", 0, 100)
    assert f(0) == 1.0 and abs(f(50) - 0.5) < 1e-12 and abs(f(100)) < 1e-12
    f = get_lr_lambda("cosine_with_restarts", 0, 100, num_cycles=2)
    assert f(0) == 1.0 and abs(f(25) - 0.5) < 1e-12 and abs(f(50) - 1.0) < 1e-12 and f(100) == 0.0
    f = get_lr_lambda("polynomial", 0, 100, lr_init=1e-3)
    assert f(0) == 1.0 and abs(f(50) - (0.5 * (1e-3 - 1e-7) + 1e-7) / 1e-3) < 1e-12 and abs(f(150) - 1e-4) < 1e-12
    with pytest.raises(ValueError):
        get_lr_lambda("exponential")
    s = LRSchedule("linear", 2e-4, 2, 6)

    class T:
        lr = None
    t, seen = T(), []
    for _ in range(6):
        s.apply(t)
        seen.append(t.lr)
        s.step()
    assert seen[0] == 0.0 and seen[2] == 2e-4 and math.isclose(seen[4], 1e-4) and s.get_last_lr()[0] == 0.0


def test_tuning_step_with_grad_clip_matches_oracle(emu_fp32):
    """config C4 (tuning_e4t.py:139-147,266-338): every UNet parameter + the encoder head train, one image expanded to the
    batch, global gradient-norm clipping at 1.0, AdamW — native flat-buffer step vs oracle + clip_grad_norm_ + torch AdamW."""
    from e4t.trainer import E4TTrainer
    r_unet, r_enc, n_unet, n_enc, text = build()
    lr = 1e-3
    tr = E4TTrainer(n_unet, n_enc, text, vae=None, lr=lr, reg_lambda=0.1, class_token_id=11, empty_prompt_ids=torch.zeros(1, 9, dtype=torch.long),
                    device=torch.device("cpu"), tuning=True, max_grad_norm=1.0)
    d = _data(1, 41)
    B = 3
    px, lat = d["px"].expand(B, -1, -1, -1).contiguous(), d["lat"].expand(B, -1, -1, -1).contiguous()
    g = torch.Generator().manual_seed(42)
    noise, t = torch.randn(B, 4, 16, 16, generator=g), torch.tensor([10, 500, 900])
    ids, pidx = d["ids"].expand(B, -1).contiguous(), d["pidx"].expand(B).contiguous()
    r_unet.requires_grad_(True)
    params = [p for p in r_enc.parameters() if p.requires_grad] + list(r_unet.parameters())
    opt = torch.optim.AdamW(params, lr=lr)
    acp = orc.ddpm_alphas_cumprod()
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([11]))[0]
        ctx0 = text(input_ids=torch.zeros(1, 9, dtype=torch.long))[0]
        emb = text.get_input_embeddings()(ids)
    loss, ld, lr_, _ = orc.e4t_losses(r_unet, r_enc, lambda inputs_embeds: text(inputs_embeds=inputs_embeds)[0], px, lat, noise, t, emb, pidx.tolist(),
                                      ctx0, class_embed, acp, reg_lambda=0.1)
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(params, 1.0)
    assert float(total) > 1.0                                  # the clip is active in this test
    before = {n: p.detach().clone() for n, p in r_unet.named_parameters()}
    opt.step()
    out = tr.train_step(px, ids, pidx, noise=noise, timesteps=t, latents=lat)
    torch.testing.assert_close(out[1], ld.detach(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out[2], lr_.detach(), rtol=1e-3, atol=1e-5)
    want = dict(r_unet.named_parameters())
    moved = 0
    for n, p in tr.unet.named_parameters():
        upd_r, upd_n = want[n].detach() - before[n], p.detach() - before[n]
        # Adam's first step is lr * sign-like: compare the update directions where the reference moved decisively
        big = upd_r.abs() > 0.5 * lr
        if big.any():
            agree = (torch.sign(upd_r[big]) == torch.sign(upd_n[big])).float().mean()
            assert float(agree) > 0.995, (n, float(agree))
            moved += 1
        assert float((p.detach() - want[n].detach()).abs().max()) <= 2.1 * lr, n
    assert moved > 200
