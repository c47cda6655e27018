"""GPU diagnostic (full-size case): error of the ViT tokens and of the two LeakyReLU inputs of the E4T encoder, native vs stock
autocast, and the number of kink sign flips of each."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
import copy
import torch
import parity_step as ps
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "full_sd14"
case = ps.cases()[name]
o = ps.build_oracle(case)
d = ps.make_data(case)
n = ps.build_native(case, o, dev)
vit_out = {}
orig = n["enc"].encode_vision
def ev(x):
    r = orig(x)
    vit_out["native"] = (r[0].detach().float().cpu(), r[1].detach().float().cpu())
    return r
n["enc"].encode_vision = ev
nat = ps.native_leg(case, n, d, dev)
h = o["enc"].clip_vision.register_forward_hook(lambda m, a, y: vit_out.__setitem__("cur", (y[0].detach().float().cpu(), y[1].detach().float().cpu())))
ref = ps.oracle_leg(case, o, d)
vit_out["oracle"] = vit_out.pop("cur")
h.remove()
# autocast copy: hook its own clip_vision
o2 = {k: (copy.deepcopy(v).to(dev) if v is not None else None) for k, v in o.items()}
h = o2["enc"].clip_vision.register_forward_hook(lambda m, a, y: vit_out.__setitem__("autocast", (y[0].detach().float().cpu(), y[1].detach().float().cpu())))
cal = ps.oracle_leg(case, o2, d, dev=dev, autocast=True) if False else None
# run the autocast leg on o2 directly (oracle_leg deep-copies; replicate minimal): use oracle_leg on `o` with a hook on the copy is not reachable -> patch deepcopy
real_deepcopy = copy.deepcopy
def dc(x, *a, **k):
    y = real_deepcopy(x, *a, **k)
    if x is o["enc"]:
        y.clip_vision.register_forward_hook(lambda m, a_, out: vit_out.__setitem__("autocast", (out[0].detach().float().cpu(), out[1].detach().float().cpu())))
    return y
ps.copy.deepcopy = dc
cal = ps.oracle_leg(case, o, d, dev=dev, autocast=True)
ps.copy.deepcopy = real_deepcopy
for who in ("native", "autocast"):
    a, b = vit_out[who], vit_out["oracle"]
    print(f"ViT {who:<9s}: pooled rel {ps.rel(a[0], b[0]):.3e}   tokens rel {ps.rel(a[1].reshape(b[1].shape), b[1]):.3e}")
for who, res in (("native", nat), ("autocast", cal)):
    for i, (x, y) in enumerate(zip(res["_kinks"], ref["_kinks"])):
        y = y.reshape(x.shape)
        fl = torch.sign(x) != torch.sign(y)
        print(f"{who:<9s} leaky input {i}: rel {ps.rel(x, y):.3e} flips {int(fl.sum())}/{x.numel()}  median|y| {float(y.abs().median()):.3e}  max |y| at flips {float(y[fl].abs().max()) if fl.any() else 0:.3e}")
for k in ("grad/e4t_encoder.first_linears.0.bias", "grad/e4t_encoder.final_linear.weight", "grad/e4t_encoder.unet_feature_embedder.0.weight", "domain_embed"):
    print(f"{k}: native {ps.rel(nat[k], ref[k]):.3e} autocast {ps.rel(cal[k], ref[k]):.3e}")
