"""CPU: the oracle against the golden vectors generated from the real reference
(tests/golden/make_golden.py imports /root/reference/e4t/weightoffsets.py)."""
import glob
import os

import pytest
import torch

import e4t_oracle as orc

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "weightoffsets_*.pt")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_weightoffsets_literal_matches_reference(path):
    blob = torch.load(path)
    m = orc.WeightOffsets(blob["row"], blob["col"]).double()
    m.load_state_dict(blob["params"])
    out = m()
    assert torch.equal(out, blob["out"]), "oracle WeightOffsets is not bit-identical to the reference class"
    out.backward(blob["upstream"])
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, blob["grads"][k]), k


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_closed_form_matches_reference(path):
    blob = torch.load(path)
    p = blob["params"]
    out = orc.weight_offsets_closed_form(p["v"], p["linear1.weight"], p["linear1.bias"], p["linear2.weight"], p["linear2.bias"],
                                         p["linear_column.weight"], p["linear_column.bias"], p["linear_row.weight"], p["linear_row.bias"])
    torch.testing.assert_close(out, blob["out"], rtol=1e-12, atol=1e-12)


def test_unet_feature_width_known_answer():
    """The reference's only stated known answer: 13 pooled maps concatenate to 10880 (unet_2d_condition.py:586)."""
    boc = (320, 640, 1280, 1280)
    assert boc[0] + sum(2 * c for c in boc) + sum(boc[:-1]) + boc[-1] == 10880
    # and the tiny oracle UNet really returns 13 maps with that channel structure
    cfg = orc.tiny_unet_config()
    unet = orc.UNet2DConditionModel(**cfg)
    x = torch.randn(1, 4, 16, 16)
    enc = unet(x, torch.tensor([10]), torch.randn(1, 5, cfg["cross_attention_dim"]), return_encoder_outputs=True)
    maps = enc["down_block_samples"]
    assert len(maps) == 13
    b = cfg["block_out_channels"]
    assert [m.shape[1] for m in maps] == [b[0], b[0], b[0], b[0], b[1], b[1], b[1], b[2], b[2], b[2], b[3], b[3], b[3]]
