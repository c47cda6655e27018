import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
d = a["g"].shape[1] // 3
rel = lambda x, y: float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30))
print("o", rel(a["o"], b["o"]), "lse", rel(a["lse"], b["lse"]))
for i, n in enumerate(("dq", "dk", "dv")):
    x, y = a["g"][:, i * d:(i + 1) * d], b["g"][:, i * d:(i + 1) * d]
    print(n, rel(x, y), "max abs diff", float((x.float() - y.float()).abs().max()), "nan", bool(torch.isnan(x.float()).any()))
    if n == "dk":
        e = (x.float() - y.float()).abs().reshape(-1, x.shape[1])
        rows = e.max(1).values
        bad = (rows > 10 * rows.median()).nonzero().flatten()
        print("  rows with large dk error:", bad.numel(), bad[:20].tolist(), "cols of worst row:", e[rows.argmax()].topk(5).indices.tolist())
