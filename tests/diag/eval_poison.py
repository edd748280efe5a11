"""Do the eval-mode forward (hc = 1: logits) and the SK solve depend on what fresh allocations contain or on what ran
before?  (diagnostic for test_two_rank_cluster_rounds_match_single_process)"""
import hashlib, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
POISON = os.environ.get("DET_POISON")
if POISON:
    _pv = float(POISON)
    _e, _el = torch.empty, torch.empty_like
    def _pe(*a, **k):
        t = _e(*a, **k)
        return t.fill_(_pv) if t.is_floating_point() and t.is_cuda else t
    def _pel(*a, **k):
        t = _el(*a, **k)
        return t.fill_(_pv) if t.is_floating_point() and t.is_cuda else t
    torch.empty, torch.empty_like = _pe, _pel
from oracle.model_ref import portable_init_
from selavi_amd import model as smodel, sk_utils
from selavi_amd.data import SyntheticAVDataset
from selavi_amd.utils import warmup_batchnorm
import tests.test_cluster_gpu as T


def h(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:10]


K = 8
ds = SyntheticAVDataset(n=192, T=4, S=32, F=40, Tp=36, n_classes=K)
m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=1)
portable_init_(m, seed=31)
m = m.cuda().train()
m.set_sync_bn(False)
loader = [(torch.stack([ds[i][0] for i in range(b, b + 16)]), torch.stack([ds[i][1] for i in range(b, b + 16)])) for b in range(0, 64, 16)]
warmup_batchnorm(T.Args(), m, loader, batches=4)
m.eval()


def logits(lo):
    v = torch.stack([ds[i][0] for i in range(lo, lo + 32)]).cuda()
    a = torch.stack([ds[i][1] for i in range(lo, lo + 32)]).cuda()
    with torch.no_grad():
        fv, fa = m(v, a)
    return fv, fa


order = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [96, 0, 32, 96, 128, 96]
for lo in order:
    fv, fa = logits(lo)
    print(f"poison {POISON} batch {lo:3d}: logits_v {h(fv)} logits_a {h(fa)} finite {bool(torch.isfinite(fv).all())}", flush=True)
# the solve on one fixed matrix, repeated
banks = [logits(lo) for lo in range(0, 192, 32)]
lv, la = torch.cat([b[0] for b in banks]), torch.cat([b[1] for b in banks])
for rep in range(3):
    PS = sk_utils.head_probabilities(lv, la)
    cost, L = sk_utils.optimize_L_sk_gpu(T.Args(distribution='default'), PS, 0, None)
    print(f"poison {POISON} SK rep {rep}: PS {h(sk_utils.head_probabilities(lv, la))} cost {cost:.17g} L {h(L)} iters {sk_utils.optimize_L_sk_gpu.last_info['iters']}", flush=True)
