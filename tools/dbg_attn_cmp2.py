import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
B, H, S, DH = (int(x) for x in sys.argv[3:7])
d = H * DH
x, y = a["g"][:, d:2 * d].float(), b["g"][:, d:2 * d].float()
e = (x - y).reshape(B, S // 128, 128, H, DH).abs().amax(dim=(2, 4))      # [B, kb, H]
ref = y.abs().max()
bad = (e > 0.05 * ref)
print("bad (b, kb, h) workgroups:", int(bad.sum()), "of", bad.numel())
gx, gy = S // 128, H
nwg = gx * gy * B
lst = []
for bb in range(B):
    for kb in range(S // 128):
        for hh in range(H):
            if bad[bb, kb, hh]:
                lin2 = (bb * gy + hh) * gx + kb          # position in the re-dealt raster
                q = nwg >> 3
                xcd, pos = lin2 // q, lin2 % q
                lst.append((xcd, pos, bb, hh, kb))
lst.sort()
print("as (xcd, position in the XCD's chunk, b, h, kb):", lst[:60])
import collections
print("positions histogram (pos // 8):", sorted(collections.Counter(p // 8 for _, p, _, _, _ in lst).items()))
# which key rows inside a bad workgroup are wrong (per wave = 32 keys)?
ew = (x - y).reshape(B, S // 128, 4, 32, H, DH).abs().amax(dim=(3, 5))    # [B, kb, wave, H]
print("bad by wave index:", [(int((ew[:, :, w, :] > 0.05 * ref).sum())) for w in range(4)])
