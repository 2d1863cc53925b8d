# Probe of the folded-LayerNorm fp32 epilogue (round 4): which operand of out = aff.x * (alpha acc) + aff.y * colsum + bias + residual
# goes wrong?  Row affines of five kinds against the plain kernel + host arithmetic; see DESIGN.md "hardware facts" (the
# v_pk_mul_f32 op_sel fault) and profiles/r4/ln_fold_opsel_fault.txt.  DBG_SO=<path> loads another build of the library.
import torch, sys, os
sys.path.insert(0, ".")
import lemon_amd._lib as L
if os.environ.get("DBG_SO"): L.SO_PATH = os.environ["DBG_SO"]
from lemon_amd import ops
m,k,n = 2500,768,3072
g = torch.Generator().manual_seed(1)
x = torch.randn(m, k, generator=g); w = 0.03*torch.randn(n,k,generator=g); b = 0.1*torch.randn(n,generator=g)
res = torch.randn(m, n, generator=g)
xc, wc, bc, rc = x.cuda(), w.cuda(), b.cuda(), res.cuda()
xt, _ = ops.rowstats_t(xc, 1e-5)
ws = ops.weight_scale_f16x3(wc); wt = ops.pack_weight_t(wc, ws)
cs = torch.randn(n, generator=g).cuda()
g0 = ops.linear_t(xt, wt, m, n, k, None, residual=None, alpha=1.0/ws)          # alpha v
def probe(name, aff):
    tot=0
    for rep in range(4):
        g1 = ops.linear_t_ln(xt, wt, m, n, k, bc, residual=rc, alpha=1.0/ws, row_aff=aff, colsum=cs)
        want = aff[:,0:1]*g0 + aff[:,1:2]*cs[None,:] + bc[None,:] + rc
        d = (g1-want).abs(); bad = d > 1e-3*(1+want.abs())
        tot += int(bad.sum())
        if rep==0 and bad.any():
            idx=bad.nonzero(); print("   rows%8", sorted(set((idx[:,0]%8).tolist())), "cols%4", sorted(set((idx[:,1]%4).tolist())), "max", float(d.max()))
    print(name, "bad total", tot)
aff = torch.zeros(m,2).cuda()
aff[:,0]=1; probe("identity", aff)
aff[:,1]=0.5; probe("const y", aff)
aff[:,1]=torch.randn(m).cuda(); probe("rand y", aff)
aff[:,1]=0; aff[:,0]=1+0.5*torch.rand(m).cuda(); probe("rand x only", aff)
aff[:,1]=torch.randn(m).cuda(); probe("rand x,y", aff)
