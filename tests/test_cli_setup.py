"""CPU: the set-up code of the three scripts (everything between the argument parser and the loop) on the op emulation with
toy models and the offline tokenizer — the glue the reference has at pretrain_e4t.py:233-259,354-361,561-583 and
tuning_e4t.py:96-147,240-265: placeholder token + embedding resize, class-token id from --domain_class_token, tokenizer("")
for the E4T encoder pass, the template lists, strict checkpoint loading, pretrained_args hand-over to inference."""
import argparse
import json
import os
import sys

import pytest
import torch

from test_unet_host_logic import emu_fp32  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CPU = torch.device("cpu")


def pretrain_args(**kw):
    d = dict(pretrained_model_name_or_path=None, unet_variant="tiny-test", seed=3, unfreeze_clip_vision=False, synthetic_data=True,
             placeholder_token="*s", domain_class_token="art", prompt_template="art", enable_xformers_memory_efficient_attention=True,
             learning_rate=1e-6, scale_lr=True, gradient_accumulation_steps=2, train_batch_size=4, domain_embed_scale=0.1, reg_lambda=0.01,
             prediction_type="epsilon", per_rank_seed=False)
    d.update(kw)
    return argparse.Namespace(**d)


def tuning_args(**kw):
    d = dict(pretrained_model_name_or_path=None, unet_variant="tiny-test", seed=3, unfreeze_clip_vision=False, synthetic_data=True,
             prompt_template=None, enable_xformers_memory_efficient_attention=False, learning_rate=1.6e-5, scale_lr=False,
             gradient_accumulation_steps=1, train_batch_size=2, domain_embed_scale=0.1, reg_lambda=1e-4, max_grad_norm=1.0, train_text_encoder=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_template_lists_are_the_references():
    from e4t import cli_common as cc
    assert (len(cc.templates), len(cc.face_templates), len(cc.art_templates)) == (10, 16, 12)        # pretrain_e4t.py:36-62
    assert cc.resolve_prompt_templates("face")[-1] == "photo realistic portrait of {placeholder_token}"
    assert cc.resolve_prompt_templates("a drawing of {placeholder_token}") == ["a drawing of {placeholder_token}"]
    with pytest.raises(AssertionError):
        cc.resolve_prompt_templates("a drawing")
    ref = os.path.join("/root/reference", "pretrain_e4t.py")
    if os.path.exists(ref):                      # build container only: the lists are the reference's, string for string
        ns = {}
        src = open(ref).read()
        exec(src[src.index("templates = ["):src.index("def parse_args")], ns)
        assert ns["templates"] == cc.templates and ns["face_templates"] == cc.face_templates and ns["art_templates"] == cc.art_templates


def test_pretrain_setup_conditions_like_the_reference(emu_fp32):
    import pretrain_e4t
    st = pretrain_e4t.setup(pretrain_args(), CPU, world=2, rank=0)
    tok, text, tr = st["tokenizer"], st["text"], st["trainer"]
    emb = text.get_input_embeddings().weight
    assert len(tok) == 101 and emb.shape[0] == 101 and st["placeholder_token_id"] == 100           # :253-259: one row added for "*s"
    assert st["class_token_id"] == tok("art", add_special_tokens=False).input_ids[0, 0] == 11       # :561-563, not a constant
    assert st["empty_ids"].tolist() == [[1] + [2] * 8]                                               # tokenizer(""): BOS + EOS padding
    torch.testing.assert_close(tr.class_embed, emb[11].detach())
    torch.testing.assert_close(tr.ctx_for_e4t, text(input_ids=st["empty_ids"])[0].detach())          # :565-583, not text(zeros)
    assert st["lr"] == 1e-6 * 2 * 4 * 2                                                              # :354-357: ga * batch * processes
    assert len(st["prompt_templates"]) == 12
    ids, pidx = st["prompts"](4)
    assert ids.shape == (4, 9) and all(ids[i, pidx[i]] == 100 for i in range(4)) and int(ids.max()) == 100
    assert type(st["unet"].down_blocks[0].attentions[0].transformer_blocks[0].attn1.processor).__name__ == "HipAttnProcessor"
    # a second placeholder registration must fail like the reference's (ValueError, :255-256)
    from e4t import cli_common as cc
    with pytest.raises(ValueError, match="already contains"):
        cc.add_placeholder_token(tok, text, "*s")
    # the class token must be a single token
    with pytest.raises(AssertionError, match="single token"):
        cc.conditioning_ids(tok, "oil painting")
    # one optimiser step through the script's own batch generator runs
    data = pretrain_e4t.synthetic_batches(pretrain_args(resolution=64), CPU, 0, 1, st["prompts"])
    px, ids, pidx = next(data)
    out = tr.train_step(px, ids, pidx, latents=torch.randn(4, 4, 16, 16) * 0.18215)
    assert all(torch.isfinite(o) for o in out)


def _write_base(tmp_path, corrupt=False):
    """a local 'Stable Diffusion checkpoint directory' of the toy architecture: plain state dicts, 100-row token table"""
    from e4t import builders
    from e4t.vae import VAEDecoder
    unet, enc, text, vae = builders.build_models(CPU, "tiny-test", seed=11)
    base = tmp_path / "sd-base"
    base.mkdir()
    usd = {k: v for k, v in unet.state_dict().items() if "wo" not in k}          # a stock UNet has no weight offsets
    if corrupt:
        usd["conv_inn.weight"] = usd.pop("conv_in.weight")
    torch.save(usd, base / "unet.pt")
    torch.save(text.state_dict(), base / "text_encoder.pt")
    dec = VAEDecoder(block_out_channels=(64, 64))
    torch.save({**vae.state_dict(), **dec.state_dict()}, base / "vae.pt")         # AutoencoderKL: encoder + decoder halves
    return str(base), unet, text, vae


def test_base_weights_load_strictly_then_the_table_grows(emu_fp32, tmp_path):
    import pretrain_e4t
    base, unet0, text0, vae0 = _write_base(tmp_path)
    st = pretrain_e4t.setup(pretrain_args(pretrained_model_name_or_path=base, seed=5), CPU)
    torch.testing.assert_close(st["unet"].conv_in.weight, unet0.conv_in.weight)                     # not the seed-5 random init
    torch.testing.assert_close(st["vae"].quant_conv.weight, vae0.quant_conv.weight)
    emb = st["text"].get_input_embeddings().weight
    assert emb.shape[0] == 101                                                                      # 100-row checkpoint loaded, THEN resized
    torch.testing.assert_close(emb[:100], text0.get_input_embeddings().weight)
    bad, _, _, _ = _write_base(tmp_path / "x" if (tmp_path / "x").mkdir() is None else tmp_path, corrupt=True)
    with pytest.raises(RuntimeError, match="conv_in"):
        pretrain_e4t.setup(pretrain_args(pretrained_model_name_or_path=bad), CPU)
    with pytest.raises(FileNotFoundError, match="tokenizer"):                                        # real data needs the real tokenizer
        pretrain_e4t.setup(pretrain_args(pretrained_model_name_or_path=base, synthetic_data=False), CPU)


def test_tuning_setup_reads_the_pretrained_run(emu_fp32, tmp_path):
    import pretrain_e4t
    import tuning_e4t
    from e4t.utils import save_config, save_e4t_encoder, save_e4t_unet
    base, _, _, _ = _write_base(tmp_path)
    pre = pretrain_e4t.setup(pretrain_args(pretrained_model_name_or_path=base, placeholder_token="*x", domain_class_token="photo", prompt_template="normal"), CPU)
    with torch.no_grad():
        for n, p in pre["unet"].named_parameters():
            if n.endswith("wo_q.v"):
                p.fill_(3.25)
    run = str(tmp_path / "run" / "100")
    save_config(vars(pretrain_args(pretrained_model_name_or_path=base, placeholder_token="*x", domain_class_token="photo", prompt_template="normal")), run)
    save_e4t_unet(pre["unet"], run)
    save_e4t_encoder(pre["enc"], run)
    st = tuning_e4t.setup(tuning_args(pretrained_model_name_or_path=run, synthetic_data=True), CPU)
    tok = st["tokenizer"]
    assert st["pretrained_args"].placeholder_token == "*x" and tok.convert_tokens_to_ids("*x") == st["placeholder_token_id"] == 100
    assert st["class_token_id"] == tok("photo", add_special_tokens=False).input_ids[0, 0] == 6     # from the PRE-TRAINED args (:249)
    assert len(st["prompt_templates"]) == 10                                                        # prompt_template None -> pretrained "normal"
    assert float(st["unet"].down_blocks[0].attentions[0].transformer_blocks[0].attn1.wo_q.v) == 3.25   # weight_offsets.pt on top of the base
    torch.testing.assert_close(st["enc"].final_linear.weight, pre["enc"].final_linear.weight)
    tr = st["trainer"]
    assert tr.tuning and tr.max_grad_norm == 1.0 and not tr.text_trainable
    assert all(p.requires_grad for p in st["unet"].parameters())
    ids, pidx = st["prompts"](2)
    out = tr.train_step(torch.rand(2, 3, 64, 64) * 2 - 1, ids, pidx, latents=torch.randn(2, 4, 16, 16) * 0.18215)
    assert all(torch.isfinite(o) for o in out)
    # what tuning saves lets inference.py find the base model: config.json carries pretrained_args (:224-227)
    cfg = dict(vars(tuning_args()), pretrained_args=dict(st["pretrained_args"]))
    save_config(cfg, str(tmp_path / "tuned"))
    got = json.load(open(tmp_path / "tuned" / "config.json"))
    assert got["pretrained_args"]["pretrained_model_name_or_path"] == base and got["pretrained_args"]["placeholder_token"] == "*x"
    with pytest.raises(SystemExit):
        tuning_e4t.setup(tuning_args(synthetic_data=False), CPU)


def test_trainable_text_encoder_trains_natively(emu_fp32):
    """tuning_e4t.py --train_text_encoder (:145-146): the text encoder's parameters join the flat buffer, its gradients come from
    the same kernels (no stock-torch path), class embedding and E4T context are re-evaluated every step."""
    import tuning_e4t
    st = tuning_e4t.setup(tuning_args(train_text_encoder=True), CPU)
    tr, text = st["trainer"], st["text"]
    assert tr.text_trainable and all(p.requires_grad for p in text.parameters())
    n_text = sum(p.numel() for p in text.parameters())
    assert tr.flat.numel >= n_text + sum(p.numel() for p in st["unet"].parameters())
    before = {n: p.detach().clone() for n, p in text.named_parameters()}
    ctx_before = tr.ctx_for_e4t.clone()
    ids, pidx = st["prompts"](2)
    px, lat = torch.rand(2, 3, 64, 64) * 2 - 1, torch.randn(2, 4, 16, 16) * 0.18215
    tr.lr = 1e-3
    tr.train_step(px, ids, pidx, latents=lat)
    moved = [n for n, p in text.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert any("q_proj.weight" in n for n in moved) and any("fc1.bias" in n for n in moved) and any("layer_norm1.weight" in n for n in moved)
    assert any("token_embedding" in n for n in moved) and any("position_embedding" in n for n in moved)
    tr.train_step(px, ids, pidx, latents=lat)
    assert not torch.equal(tr.ctx_for_e4t, ctx_before)                                             # re-evaluated with the updated weights


def test_checked_load_ignores_legacy_position_ids_but_nothing_else(tmp_path):
    """CLIPTextModel state dicts written by transformers < 4.31 carry the persistent buffer `...embeddings.position_ids`
    (an arange): real SD text-encoder dumps must load; any other unexpected key still raises."""
    from e4t import cli_common as cc
    lin = torch.nn.Linear(4, 3)
    sd = dict(lin.state_dict())
    sd["text_model.embeddings.position_ids"] = torch.arange(77)[None]
    torch.save(sd, tmp_path / "ok.pt")
    cc.checked_load(torch.nn.Linear(4, 3), str(tmp_path / "ok.pt"))
    sd["text_model.embeddings.something_else"] = torch.zeros(1)
    torch.save(sd, tmp_path / "bad.pt")
    with pytest.raises(RuntimeError, match="something_else"):
        cc.checked_load(torch.nn.Linear(4, 3), str(tmp_path / "bad.pt"))
