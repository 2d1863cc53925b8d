import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import lemon_amd
from tests.synth import unit_rows
for (nq,n,d,k) in [(130,256,40,10)]:
    rng = np.random.default_rng(nq * 7 + n * 3 + d + k)
    X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
    idx = lemon_amd.IndexFlatIP(d); idx.set_algo(2); idx.add(torch.from_numpy(X).cuda())
    D, I = idx.search(torch.from_numpy(Q).cuda(), k); torch.cuda.synchronize()
