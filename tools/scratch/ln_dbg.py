import torch, sys
sys.path.insert(0, ".")
from lemon_amd import ops
def run(m,k,n,mos):
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g) * (0.5 + torch.rand(m, 1, generator=g) * 4) + mos * torch.randn(m, 1, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(k, generator=g), 0.2 * torch.randn(k, generator=g)
    w, b = 0.03 * torch.randn(n, k, generator=g), 0.1 * torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    eps=1e-5
    xd = x.double()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + eps) * gamma.double() + beta.double()
    want = ln @ w.double().t() + b.double()
    xc, wc, bc, gc, bec, rc = (t.cuda() for t in (x, w, b, gamma, beta, res))
    xt, aff = ops.rowstats_t(xc, eps)
    wt, a, cs, bp = ops.fold_layernorm_weight(wc, bc, gc, bec, 1.0)
    got = ops.linear_t_ln(xt, wt, m, n, k, bp, residual=rc, alpha=a, row_aff=aff, colsum=cs).cpu().double() - res.double()
    err = (got - want).abs()
    bad = err > 1e-3
    print(m,k,n, "max err", float(err.max()), "bad", int(bad.sum()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print(" bad rows", rows[:10].tolist(), "...", rows[-5:].tolist(), len(rows)); print(" bad cols", cols[:10].tolist(), "...", cols[-5:].tolist(), len(cols))
        r0=int(rows[0]); c0 = int(cols[0]); print(" sample got/want", got[r0, c0].item(), want[r0,c0].item())
    # aff check
    affd = aff.cpu().double(); rstd = 1/torch.sqrt(xd.var(1,unbiased=False)+eps)
    print(" aff err", float((affd[:,0]-rstd).abs().max()), float((affd[:,1]+xd.mean(1)*rstd).abs().max()))
for a in [(300,128,512,0.0),(1000,768,2304,3.0),(2500,768,3072,0.3),(2500,768,768,0.3),(2500,768,2304,0.3),(2048,768,3072,0.3),(2100,768,1024,0.3)]:
    run(*a)
