import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import lemon_amd
from oracle import oracle as o
from tests.synth import unit_rows
for (nq, n, d, k) in [(130,128,40,10),(130,256,40,10),(130,384,40,10),(130,1000,40,10),(130,1000,40,1),(4,1000,40,10),(130,1000,512,10),(130,1000,768,10)]:
    rng = np.random.default_rng(nq * 7 + n * 3 + d + k)
    X, Q = unit_rows(rng, n, d), unit_rows(rng, nq, d)
    idx = lemon_amd.IndexFlatIP(d); idx.set_algo(2); idx.add(torch.from_numpy(X).cuda())
    D, I = idx.search(torch.from_numpy(Q).cuda(), k); D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = o.knn("ip", X, Q, k)
    bad = np.where((I != Ir).any(1))[0]
    nmiss = (I == -1).sum()
    print((nq,n,d,k), "bad queries", len(bad), "minus1", nmiss, "first bad", bad[:8])
