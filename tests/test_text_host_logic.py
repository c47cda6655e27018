"""CPU: the native CLIP text encoder (e4t/text.py: fused q|k|v GEMM, causal fused attention, residual epilogues) against its
stock-torch twin (tests/torch_twins.py = the HF CLIPTextModel semantics the reference uses through modeling_clip.py) — forward and the
gradient w.r.t. inputs_embeds, both activations, through the fp32 op emulation."""
import pytest
import torch

from test_unet_host_logic import emu_fp32  # noqa: F401


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_encoder_matches_torch_twin(emu_fp32, act):
    from torch_twins import CLIPTextModel as TorchText
    from e4t.text import CLIPTextModel as NativeText
    cfg = dict(vocab_size=120, hidden_size=64, num_layers=3, num_heads=2, intermediate_size=160, max_len=77, act=act)
    torch.manual_seed(0)
    ref = TorchText(**cfg).requires_grad_(False)
    nat = NativeText(**cfg).requires_grad_(False)
    nat.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(1)
    B, S = 3, 77
    ids = torch.randint(0, 120, (B, S), generator=g)
    torch.testing.assert_close(nat(input_ids=ids)[0], ref(input_ids=ids)[0], rtol=2e-4, atol=2e-5)
    e1 = (torch.randn(B, S, 64, generator=g) * 0.5).requires_grad_(True)
    e2 = e1.detach().clone().requires_grad_(True)
    y1, y2 = nat(inputs_embeds=e1)[0], ref(inputs_embeds=e2)[0]
    torch.testing.assert_close(y1, y2, rtol=2e-4, atol=2e-5)
    w = torch.randn(B, S, 64, generator=g)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    torch.testing.assert_close(e1.grad, e2.grad, rtol=5e-4, atol=5e-5)
    # causal: the output at position t must not depend on later tokens
    e3 = e1.detach().clone()
    e3[:, 40:] += 1.0
    torch.testing.assert_close(nat(inputs_embeds=e3)[0][:, :40], y1.detach()[:, :40], rtol=1e-5, atol=1e-6)


def test_trainable_text_encoder_gradients_match_torch_twin(emu_fp32):
    """tuning_e4t.py --train_text_encoder: every parameter gradient of the native text encoder (fused q|k|v assembled by a
    differentiable cat, TN weight-gradient GEMMs, LayerNorm / bias gradients from the kernels) vs autograd on the torch twin."""
    from torch_twins import CLIPTextModel as TorchText
    from e4t.text import CLIPTextModel as NativeText
    cfg = dict(vocab_size=50, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=128, max_len=9, act="quick_gelu")
    torch.manual_seed(0)
    ref, nat = TorchText(**cfg), NativeText(**cfg)
    nat.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 50, (3, 9), generator=g)
    w = torch.randn(3, 9, 64, generator=g)
    (ref(input_ids=ids)[0] * w).sum().backward()
    (nat(input_ids=ids)[0] * w).sum().backward()
    want = dict(ref.named_parameters())
    n = 0
    for name, p in nat.named_parameters():
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad, want[name].grad, rtol=2e-3, atol=2e-5, msg=lambda m, name=name: f"{name}: {m}")
        n += 1
    assert n == len(want) == 2 + 2 * 16 + 2
    # a second forward after an in-place weight update sees the new weights (the fused copy is rebuilt every forward)
    with torch.no_grad():
        for m in (ref, nat):
            m.text_model.encoder.layers[0].self_attn.q_proj.weight.mul_(1.5)
    torch.testing.assert_close(nat(input_ids=ids)[0], ref(input_ids=ids)[0], rtol=2e-4, atol=2e-5)


def test_text_encoder_matches_installed_transformers_clip():
    """[3P] leaf pinned against the real third-party code that IS installed: transformers' CLIPTextModel (5.x: no `text_model.`
    prefix in its keys and no inputs_embeds argument — the reason the reference's modeling_clip.py patch cannot import here).
    Same weights by key -> same last_hidden_state for the torch twin, the oracle and (through the emulation) the native class;
    the inputs_embeds entry point equals the input_ids one."""
    transformers = pytest.importorskip("transformers")
    import e4t_oracle as orc
    from torch_twins import CLIPTextModel as Twin
    for act, heads in (("quick_gelu", 2), ("gelu", 4)):
        cfg = transformers.CLIPTextConfig(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=heads,
                                          max_position_embeddings=9, hidden_act=act, bos_token_id=1, eos_token_id=2)
        torch.manual_seed(0)
        hf = transformers.CLIPTextModel(cfg).eval()
        sd = {("" if k.startswith("text_model.") else "text_model.") + k: v for k, v in hf.state_dict().items()}
        ids = torch.randint(3, 99, (2, 9), generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            want = hf(input_ids=ids).last_hidden_state
        mine = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=heads, intermediate_size=128, max_len=9, act=act)
        twin = Twin(**mine)
        missing, unexpected = twin.load_state_dict(sd, strict=False)
        assert not missing and all("position_ids" in k for k in unexpected)
        with torch.no_grad():
            got = twin(input_ids=ids)[0]
            via_embeds = twin(inputs_embeds=twin.get_input_embeddings()(ids))[0]
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(via_embeds, got, rtol=0, atol=0)
        o = orc.CLIPTextModel(vocab=100, width=64, layers=2, heads=heads, mlp=128, max_len=9, act=act)
        flat = {}
        for k, v in hf.state_dict().items():          # the oracle keeps a flat naming: layers.i.{q_proj,...,fc1,fc2,layer_norm*}
            k = k.replace("text_model.", "").replace("embeddings.", "").replace("encoder.layers.", "layers.").replace("self_attn.", "").replace("mlp.", "")
            if "position_ids" not in k:
                flat[k] = v
        o.load_state_dict(flat)
        with torch.no_grad():
            torch.testing.assert_close(o(input_ids=ids), want, rtol=1e-5, atol=1e-5)


def test_prepared_linear_recasts_in_place_after_an_optimiser_step(emu_fp32):
    """A trainable weight is re-cast once per optimiser step (weights epoch bumped by the raw-pointer AdamW): the compute copies
    and the descriptor table are kept — same storage, new values — and an unchanged weight is not re-cast at all."""
    from e4t import functional as Fn
    from e4t import ops
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(24, 16))
    prep = Fn.PreparedLinear(w)
    a, aT = prep.get()
    table = prep._table
    pa, paT = a.data_ptr(), aT.data_ptr()
    torch.testing.assert_close(a.float()[:, :16], w.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(aT.float()[:, :24], w.detach().t(), rtol=1e-2, atol=1e-2)
    b, _ = prep.get()                                    # nothing changed: no re-cast, same objects
    assert b is a and prep._table is table
    with torch.no_grad():
        w.data.mul_(2.0)                                 # what the fused AdamW does: raw write, then bump the epoch
    ops.bump_weights_epoch()
    c, cT = prep.get()
    assert c.data_ptr() == pa and cT.data_ptr() == paT and prep._table is table
    torch.testing.assert_close(c.float()[:, :16], w.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(cT.float()[:, :24], w.detach().t(), rtol=1e-2, atol=1e-2)
    # a frozen weight ignores the epoch
    f = torch.nn.Parameter(torch.randn(8, 8), requires_grad=False)
    pf = Fn.PreparedLinear(f)
    x, _ = pf.get()
    key = pf.key
    ops.bump_weights_epoch()
    y, _ = pf.get()
    assert y is x and pf.key == key
