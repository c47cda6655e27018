"""Per-shape time of every GEMM / conv launch in one training step (events around each launch).
usage: shape_profile.py [sd14|sd21] [B]      (sd21 = BASELINE configs[4]: 768 px, v-prediction)"""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R]
import torch
from e4t import ops
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
hip = ops.backend()
MODEL = sys.argv[1] if len(sys.argv) > 1 else "sd14"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
RES = 768 if MODEL == "sd21" else 512
unet, enc, text, vae = bench.build_models(dev, MODEL, 0)
from e4t.trainer import E4TTrainer
kw = dict(prediction_type="v_prediction", empty_prompt_ids=torch.tensor([[49406] + [49407] * 76], device=dev)) if MODEL == "sd21" else {}
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, device=dev, **kw)
def batch(s):
    g = torch.Generator(device=dev); g.manual_seed(s)
    return (torch.rand((B, 3, RES, RES), generator=g, device=dev) * 2 - 1, torch.randint(0, 49000, (B, 77), generator=g, device=dev),
            torch.randint(1, 20, (B,), generator=g, device=dev))
for s in range(2): tr.train_step(*batch(s))
# monkeypatch _timed to key by shape
orig_gemm, orig_conv = hip.gemm, hip.conv3x3
rec = []
def gemm(a, b, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_gemm(a, b, **kw); e1.record()
    K = a.shape[-1] + (kw["a2"].shape[-1] if kw.get("a2") is not None else 0)
    nb = a.shape[0] if a.dim() == 3 else 1
    rec.append((f"gemm M{a.shape[-2]} N{b.shape[-2]} K{K} nb{nb} {'f32' if (kw.get('out') is not None and kw['out'].dtype==torch.float32) or kw.get('out_dtype')==torch.float32 else 'bf16'}", 2.0*a.shape[-2]*b.shape[-2]*K*nb, e0, e1)); return y
def conv(x, w, Bn, Hin, Win, Hout, Wout, mode, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_conv(x, w, Bn, Hin, Win, Hout, Wout, mode, **kw); e1.record()
    rec.append((f"conv m{mode} {Hin}x{Win} {x.shape[1]}->{w.shape[0]}", 2.0*Bn*Hout*Wout*w.shape[0]*w.shape[1], e0, e1)); return y
hip.gemm, hip.conv3x3 = gemm, conv
tr.train_step(*batch(5)); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k, fl, e0, e1 in rec:
    a = agg[k]; a[0] += fl; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(v[1] for v in agg.values())
print(f"total gemm+conv ms {tot:.1f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:110]:
    print(f"{v[1]:7.2f} ms {v[2]:4d}x  {v[0]/v[1]/1e9:7.1f} TF  {k}")
