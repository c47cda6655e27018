"""CPU: checkpoint helpers with the reference's names (e4t/utils.py): round trip of weight_offsets.pt / encoder.pt /
config.json through save_* and load_*, strictness on missing / unexpected keys."""
import json
import os

import pytest
import torch

import e4t_oracle as orc
from test_encoder_host_logic import BOC


def test_checkpoint_round_trip(tmp_path):
    from e4t.models.unet_2d_condition import UNet2DConditionModel
    from e4t.utils import (AttributeDict, load_config_from_pretrained, load_e4t_encoder, load_e4t_unet, save_config, save_e4t_encoder,
                           save_e4t_unet, weight_offset_state_dict)
    from e4t.encoder import E4TEncoder
    cfg = orc.tiny_unet_config(ctx_dim=64)
    torch.manual_seed(0)
    unet = UNet2DConditionModel(**cfg)
    enc = E4TEncoder(word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3)
    base, out = tmp_path / "base", tmp_path / "run" / "10"
    os.makedirs(base)
    torch.save({k: v for k, v in unet.state_dict().items() if "wo" not in k}, base / "unet.pt")
    (base / "unet_config.json").write_text(json.dumps(dict(cfg, _class_name="UNet2DConditionModel")))
    save_config(dict(pretrained_model_name_or_path=str(base), placeholder_token="*s", pretrained_args=None), str(out))
    save_e4t_unet(unet, str(out))
    save_e4t_encoder(enc, str(out))
    assert sorted(os.listdir(out)) == ["config.json", "encoder.pt", "weight_offsets.pt"]
    wo = torch.load(out / "weight_offsets.pt")
    assert wo.keys() == weight_offset_state_dict(unet).keys() and all("wo" in k for k in wo) and len(wo) == 9 * 96          # 96 WeightOffsets instances x 9 tensors
    c = load_config_from_pretrained(str(out))
    assert isinstance(c, AttributeDict) and c.placeholder_token == "*s" and c.not_there is None
    unet2 = load_e4t_unet(ckpt_path=str(out / "weight_offsets.pt"))               # base dir comes from config.json
    for (k, a), (_, b) in zip(unet.state_dict().items(), unet2.state_dict().items()):
        assert torch.equal(a, b), k
    enc2 = load_e4t_encoder(ckpt_path=str(out), word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3)
    assert all(torch.equal(a, b) for a, b in zip(enc.state_dict().values(), enc2.state_dict().values()))
    # strictness, as the reference: unexpected keys always raise, missing ones when a checkpoint was given
    bad = dict(wo)
    bad["down_blocks.0.bogus.wo_q.v"] = torch.zeros(1)
    torch.save(bad, out / "weight_offsets.pt")
    with pytest.raises(RuntimeError, match="unexpected keys"):
        load_e4t_unet(ckpt_path=str(out / "weight_offsets.pt"))
    sd = enc.state_dict()
    sd.pop("final_linear.bias")
    torch.save(sd, out / "encoder.pt")
    with pytest.raises(RuntimeError, match="missing keys"):
        load_e4t_encoder(ckpt_path=str(out), word_embedding_dim=64, block_out_channels=BOC, arch="ViT-tiny-test", n_odd_layers=3)
    with pytest.raises(FileNotFoundError):
        load_config_from_pretrained("e4t-diffusion-ffhq-celebahq-v1")              # hub names need a network


def test_image_helpers(tmp_path):
    import numpy as np
    from PIL import Image
    from e4t.utils import image_grid, load_image
    a = np.random.default_rng(0).integers(0, 256, (96, 128, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "a.png")
    im = load_image(str(tmp_path / "a.png"))
    assert im.size == (128, 96) and im.mode == "RGB"
    sq = load_image(str(tmp_path / "a.png"), resolution=48)
    assert sq.size == (48, 48)
    want = a.reshape(48, 2, 64, 2, 3).mean((1, 3))[:, 8:56]                          # 2x box average, centre crop
    assert np.abs(np.asarray(sq).astype(float) - want).max() <= 1.0
    g = image_grid([sq] * 6, rows=2, cols=3)
    assert g.size == (144, 96)


def test_capture_guard_pauses_the_loader_and_the_garbage_collector():
    """ops.capture_guard() is what every HIP-graph capture runs under (text encoder, sampling step): while it is open the loader's
    upload path (which takes ops.capture_lock) must block, the garbage collector must be off (a collected old graph would call
    hipGraphDestroy inside the capture), and both must be restored afterwards — also when the capture raises, and when nested."""
    import gc
    import threading
    import time

    from e4t import ops

    assert gc.isenabled()
    entered = []

    def upload():                       # what data.DeviceLoader._upload does around its GPU work
        with ops.capture_lock:
            entered.append(time.monotonic())

    with ops.capture_guard():
        assert not gc.isenabled()
        t = threading.Thread(target=upload)
        t.start()
        time.sleep(0.2)
        assert not entered, "the loader's upload ran during a capture"
        with ops.capture_guard():       # re-entrant (a pipeline capture that triggers a text-encoder capture)
            assert not gc.isenabled()
        assert not gc.isenabled(), "the inner guard re-enabled the collector while the outer capture is still open"
        released = time.monotonic()
    t.join(5)
    assert entered and entered[0] >= released
    assert gc.isenabled()
    try:
        with ops.capture_guard():
            raise RuntimeError("capture failed")
    except RuntimeError:
        pass
    assert gc.isenabled() and ops.capture_lock.acquire(blocking=False)
    ops.capture_lock.release()
