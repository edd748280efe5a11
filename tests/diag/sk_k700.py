"""Sinkhorn-Knopp at K = 700 (KJ = 12 with 68 padding columns): iterations and state against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from selavi_amd import sk_utils
from oracle import sk_ref
from _synth import synth_PS
for N, K in [(257, 700), (257, 704), (257, 690), (257, 768), (300, 640), (257, 650)]:
    PS = synth_PS(N, K, 1.5, 77 + N)
    _, L_o, info_o = sk_ref.optimize_L_sk(PS)
    P = torch.from_numpy(PS).cuda()
    r = torch.full((K,), 1.0 / K, dtype=torch.float64, device="cuda")
    for mi in (1, 2, 5, 2000):
        L, ls, info = sk_utils.sinkhorn(P.clone(), r, 20, max_iter=mi)
        a = info["alpha"].cpu().numpy()
        print(N, K, "max_iter", mi, "iters", info["iters"], "oracle", info_o["iters"], "alpha finite", np.isfinite(a).all(), "alpha[:3]", a[:3], "min/max", a.min(), a.max())
