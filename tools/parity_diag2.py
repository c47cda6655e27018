"""GPU diagnostic: gradients at module boundaries (d ctx out of the UNet, d inputs_embeds out of the text encoder, d encoder output)
of the native leg vs the emulation leg vs the oracle."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import torch
import parity_step as ps
from e4t import ops
from emu_backend import EmuBackend
dev = torch.device("cuda:0")


def leg(case, o, d, emu):
    old = ops.set_backend(EmuBackend(round_bf16=True)) if emu else None
    n = ps.build_native(case, o, dev)
    got = {}
    te = n["text"]
    orig = te.forward

    def fwd(input_ids=None, inputs_embeds=None):
        out = orig(input_ids=input_ids, inputs_embeds=inputs_embeds)
        if inputs_embeds is not None and inputs_embeds.requires_grad:
            got["emb"] = inputs_embeds.detach().float().clone()
            got["ctx"] = out[0].detach().float().clone()
            inputs_embeds.register_hook(lambda g: got.__setitem__("d_emb", g.detach().float().clone()))
            out[0].register_hook(lambda g: got.__setitem__("d_ctx", g.detach().float().clone()))
        return out
    te.forward = fwd
    n["enc"].register_forward_hook(lambda m, a, y: (y.register_hook(lambda g: got.__setitem__("d_encout", g.detach().float().clone())), None)[1])
    ps.native_leg(case, n, d, dev)
    if emu:
        ops.set_backend(old)
    return {k: v.cpu() for k, v in got.items()}


for name in sys.argv[1:]:
    case = ps.cases()[name]
    o = ps.build_oracle(case)
    d = ps.make_data(case)
    a, b = leg(case, o, d, False), leg(case, o, d, True)
    # oracle boundary gradients
    import e4t_oracle as orc
    got = {}
    text = o["text"]
    for m in (o["unet"], o["enc"]):
        for p in m.parameters():
            p.grad = None
    acp = orc.ddpm_alphas_cumprod()
    with torch.no_grad():
        class_embed = text.get_input_embeddings()(torch.tensor([case.class_id]))[0]
        ctx0 = text(input_ids=d["empty_ids"])
        emb = text.get_input_embeddings()(d["ids"])

    def tfn(inputs_embeds):
        out = text(inputs_embeds=inputs_embeds)
        inputs_embeds.register_hook(lambda g: got.__setitem__("d_emb", g.detach().clone()))
        out.register_hook(lambda g: got.__setitem__("d_ctx", g.detach().clone()))
        got["ctx"] = out.detach().clone(); got["emb"] = inputs_embeds.detach().clone()
        return out
    h = o["enc"].register_forward_hook(lambda m, a_, y: (y.register_hook(lambda g: got.__setitem__("d_encout", g.detach().clone())), None)[1])
    loss, _, _, aux = orc.e4t_losses(o["unet"], o["enc"], tfn, d["pixels"], d["latents"], d["noise"], d["t"], emb, d["pidx"].tolist(), ctx0, class_embed, acp,
                                     reg_lambda=case.reg_lambda, prediction_type=case.prediction_type)
    loss.backward()
    h.remove()
    print(f"== {name}   (rel-L2 vs oracle)      native      emu     | norms oracle")
    for k in ("emb", "ctx", "d_ctx", "d_emb", "d_encout"):
        print(f"   {k:<10s} {ps.rel(a[k].reshape(got[k].shape), got[k]):.3e}  {ps.rel(b[k].reshape(got[k].shape), got[k]):.3e}   | {float(got[k].norm()):.3e}")
    pid = d["pidx"].tolist()
    rows = lambda t: torch.stack([t.reshape(got["d_emb"].shape)[i, j] for i, j in enumerate(pid)])
    print(f"   d_emb@placeholder rows: native {ps.rel(rows(a['d_emb']), rows(got['d_emb'])):.3e} emu {ps.rel(rows(b['d_emb']), rows(got['d_emb'])):.3e}  norm {float(rows(got['d_emb']).norm()):.3e}")
    reg = 2 * case.reg_lambda * aux["domain_embed"].detach() * 0.1
    print(f"   |d_encout| oracle {float(got['d_encout'].norm()):.3e}, of which reg part {float(reg.norm()):.3e}")
    # nature of the error of the encoder gradients: scale factor or noise?
    n = ps.build_native(case, o, dev)
    nat = ps.native_leg(case, n, d, dev)
    ref = ps.oracle_leg(case, o, d)
    for k in ref:
        if k.startswith("grad/e4t_encoder") and ref[k].numel() >= 64:
            x, y = nat[k].double().reshape(-1), ref[k].double().reshape(-1)
            cos = float((x @ y) / (x.norm() * y.norm()))
            print(f"   {k[5:]:<60s} rel {ps.rel(nat[k], ref[k]):.3e}  |nat|/|ref| {float(x.norm() / y.norm()):.4f}  cos {cos:.5f}  bestfit-scale residual {float((x - (x @ y) / (y @ y) * y).norm() / x.norm()):.3e}")
    # the LeakyReLU in front of final_linear (encoder.py:163-166): do the signs of its input agree?
    from e4t import functional as Fn
    rec = []
    orig_lr = Fn.leaky_relu
    Fn.leaky_relu = lambda x: (rec.append(x.detach().float().cpu().clone()), orig_lr(x))[1]
    n = ps.build_native(case, o, dev)
    ps.native_leg(case, n, d, dev)
    Fn.leaky_relu = orig_lr
    orec = []
    h1 = o["enc"].act.register_forward_hook(lambda m, a_, y: orec.append(a_[0].detach().clone()))
    h2 = o["enc"].unet_feature_embedder[1].register_forward_hook(lambda m, a_, y: orec.append(a_[0].detach().clone()))
    ps.oracle_leg(case, o, d)
    h1.remove(); h2.remove()
    for i, (x, y) in enumerate(zip(rec[-2:], orec[-2:])):
        y = y.reshape(x.shape)
        flips = (torch.sign(x) != torch.sign(y))
        print(f"   leaky input {i}: shape {tuple(x.shape)} rel {ps.rel(x, y):.3e}  sign flips {int(flips.sum())} of {x.numel()};  |y| at flips {y[flips].abs().tolist()[:8]}  typical |y| {float(y.abs().median()):.3e}")
