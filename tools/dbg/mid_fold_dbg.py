"""tools/dbg: where does the LN-folded consumer of the mid-size kernel differ from the reference (per-row / per-column error map)."""
import sys
sys.path.insert(0, '.')
import torch
from ldt_amd import ops
from ldt_amd._lib import EPI_BF16
bf = lambda x: x.to(torch.bfloat16).float()
M, D, N2 = 2048, 1024, 3072
g = torch.Generator().manual_seed(1)
x = torch.randn(M, D, generator=g) * 1.5 + 0.6
sc = 0.3 * torch.randn(D, generator=g); sh = 0.3 * torch.randn(D, generator=g)
w2 = bf(torch.randn(N2, D, generator=g) / D ** 0.5); b2 = torch.randn(N2, generator=g)
xs = (x * (1 + sc)).to(torch.bfloat16)
t = x.double().view(M, D // 32, 32)
stats = torch.stack([t.sum(-1).T, (t ** 2).sum(-1).T], -1).float().contiguous()      # [32][M][2]
S = (w2.double() * (1 + sc.double())).sum(1).float(); C = (w2.double() @ sh.double() + b2.double()).float()
y = ops.gemm_lnfold(xs.cuda(), w2.cuda().to(torch.bfloat16), stats.cuda(), S.cuda(), C.cuda(), EPI_BF16).float().cpu().double()
xn = x.double()
mu = xn.mean(1, keepdim=True); r = 1 / torch.sqrt(xn.var(1, unbiased=False, keepdim=True) + 1e-6)
ref = r * (xs.double() @ w2.double().T) - r * mu * S.double() + C.double()
err = (y - ref) ** 2
print("rel mse", float(err.sum() / (ref ** 2).sum()))
rowe = err.sum(1) / (ref ** 2).sum(1); cole = err.sum(0) / (ref ** 2).sum(0)
print("worst rows", torch.topk(rowe, 12)); print("worst cols", torch.topk(cole, 12))
print("rows > 1e-4:", int((rowe > 1e-4).sum()), "cols > 1e-4:", int((cole > 1e-4).sum()))
bad = (rowe > 1e-4).nonzero().flatten().tolist(); print("bad rows", bad[:64])
badc = (cole > 1e-4).nonzero().flatten().tolist(); print("bad cols", badc[:64])
acc = xs.double() @ w2.double().T
for (i, j) in [(0, 12), (0, 14), (5, 76), (130, 12), (0, 13), (16, 12)]:
    print("y[%d,%d]=%.5f ref=%.5f | r*acc+C=%.5f | -r*mu*S=%.5f | C=%.5f S=%.5f r=%.4f mu=%.4f acc=%.4f" % (
        i, j, y[i, j], ref[i, j], r[i, 0] * acc[i, j] + C[j], -r[i, 0] * mu[i, 0] * S[j], C[j], S[j], r[i, 0], mu[i, 0], acc[i, j]))
y2 = ops.gemm_lnfold(xs.cuda(), w2.cuda().to(torch.bfloat16), stats.cuda(), S.cuda(), C.cuda(), EPI_BF16).float().cpu().double()
print("second run identical:", bool(torch.equal(y, y2)), " max |y - y2| at bad cols:", float((y - y2).abs().max()))
d = (y - ref)[:16, :]
print("diff rows 0..3, cols 8..15:\n", d[:4, 8:16])
# what S / C / r / mu would explain y at the bad entries?   y = r*acc - r*mu*S' + C'  -> S' - S if only S wrong
Sp = (r[:16] * acc[:16, 12] + C[12] - y[:16, 12]) / (r[:16] * mu[:16]).squeeze(1) if False else None
Sx = ((r[:16, 0] * acc[:16, 12] + C[12] - y[:16, 12]) / (r[:16, 0] * mu[:16, 0]))
print("implied S[12] per row (true %.5f):" % S[12], Sx)
Cx = y[:16, 12] - r[:16, 0] * acc[:16, 12] + r[:16, 0] * mu[:16, 0] * S[12]
print("implied C[12] per row (true %.5f):" % C[12], Cx)
