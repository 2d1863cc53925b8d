# Probe of the EMIT epilogue's row statistics (round 4): run test_gemm_emits...'s checks on one shape with details, repeated, optionally
# with another build of the library (DBG_SO=<path>).  Used to look at a packed-arithmetic variant of the statistics that failed.
import os, sys, torch
sys.path.insert(0, ".")
import lemon_amd._lib as L
if os.environ.get("DBG_SO"): L.SO_PATH = os.environ["DBG_SO"]
from lemon_amd import ops
m, k, n = 129, 2048, 768
g = torch.Generator().manual_seed(m + k + n)
x = torch.randn(m, k, generator=g)
w, b = 0.05 * torch.randn(n, k, generator=g), 0.1 * torch.randn(n, generator=g)
res = torch.randn(m, n, generator=g) * (0.2 + 3 * torch.rand(m, 1, generator=g)) + 20.0 * torch.randn(m, 1, generator=g)
xc, wc, bc, rc = (t.cuda() for t in (x, w, b, res))
ws = ops.weight_scale_f16x3(wc); wt = ops.pack_weight_t(wc, ws)
at, _ = ops.rowstats_t(xc, 1e-5)
for rep in range(4):
    out, et, st = ops.linear_t_ln(at, wt, m, n, k, bc, residual=rc, alpha=1.0 / ws, emit=True)
    od = out.cpu().double()
    stc = st.cpu().double()                       # [m, n/128, 2]
    parts = od.view(m, n // 128, 128)
    pm, pm2 = parts.mean(2), ((parts - parts.mean(2, keepdim=True)) ** 2).sum(2)
    em = (stc[:, :, 0] - pm).abs(); e2 = (stc[:, :, 1] - pm2).abs() / pm2
    bad = (em > 1e-4 * (1 + pm.abs())) | (e2 > 1e-4)
    print("rep", rep, "partial mean max err", float(em.max()), "partial M2 max rel err", float(e2.max()), "bad partials", int(bad.sum()), "rows%8", sorted(set((bad.nonzero()[:, 0] % 8).tolist())))
