import torch, sys, os
sys.path.insert(0, ".")
import lemon_amd._lib as L
if os.environ.get("DBG_SO"): L.SO_PATH = os.environ["DBG_SO"]
mode = int(os.environ["DBG_MODE"])
from lemon_amd import ops
m,k,n = 2500,768,3072
g = torch.Generator().manual_seed(1)
x = torch.randn(m, k, generator=g); w = 0.03*torch.randn(n,k,generator=g)
xc, wc = x.cuda(), w.cuda()
xt, _ = ops.rowstats_t(xc, 1e-5)
ws = ops.weight_scale_f16x3(wc); wt = ops.pack_weight_t(wc, ws)
cs = torch.randn(n, generator=g).cuda()
aff = torch.zeros(m,2).cuda(); aff[:,0] = 1 + torch.rand(m).cuda(); aff[:,1] = torch.randn(m).cuda()
for rep in range(3):
    g1 = ops.linear_t_ln(xt, wt, m, n, k, None, residual=None, alpha=1.0/ws, row_aff=aff, colsum=cs)
    if mode == 4: want = aff[:,1:2]*cs[None,:]
    elif mode == 5: want = cs[None,:].expand(m,n)
    else:
        want = torch.empty(m,n).cuda(); want[:,0::2] = aff[:,1:2]; want[:,1::2] = aff[:,0:1]
    bad = (g1 != want)
    print("mode", mode, "rep", rep, "mismatch", int(bad.sum()))
    if bad.any():
        idx = bad.nonzero(); print("  rows%8", sorted(set((idx[:,0]%8).tolist())), "cols%4", sorted(set((idx[:,1]%4).tolist())))
        r,c = idx[0].tolist(); print("  sample", r, c, float(g1[r,c]), float(want[r,c]), "cs", float(cs[c]), "aff", aff[r].tolist())
        # is the wrong value some other known quantity?
        v = float(g1[r,c])
        near = (cs - v).abs().min().item(); print("  nearest cs diff", near, " nearest aff diff", (aff - v).abs().min().item())
