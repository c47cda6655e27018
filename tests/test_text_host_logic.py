"""CPU: the native CLIP text encoder (e4t/text.py: fused q|k|v GEMM, causal fused attention, residual epilogues) against its
stock-torch twin (e4t/frozen.py = the HF CLIPTextModel semantics the reference uses through modeling_clip.py) — forward and the
gradient w.r.t. inputs_embeds, both activations, through the fp32 op emulation."""
import pytest
import torch

from test_unet_host_logic import emu_fp32  # noqa: F401


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_encoder_matches_torch_twin(emu_fp32, act):
    from e4t.frozen import CLIPTextModel as TorchText
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


def test_trainable_text_encoder_falls_back_to_torch(emu_fp32):
    from e4t.text import CLIPTextModel as NativeText
    cfg = dict(vocab_size=50, hidden_size=32, num_layers=1, num_heads=2, intermediate_size=64, max_len=9, act="quick_gelu")
    m = NativeText(**cfg)          # parameters require grad -> torch path, weight gradients exist
    ids = torch.randint(0, 50, (2, 9))
    m(input_ids=ids)[0].sum().backward()
    assert m.text_model.encoder.layers[0].mlp.fc1.weight.grad is not None
