"""GPU parity tests of the Sinkhorn-Knopp path: HIP (through the C ABI) vs the oracle and the
golden vectors of the executed reference.  Labels must be BIT-EXACT."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import sk_ref
from tests._synth import synth_PS, synth_logits

pytestmark = pytest.mark.gpu


class Args:
    def __init__(self, **kw):
        self.distribution, self.dist, self.diff_dist_every = 'default', None, False
        self.diff_dist_per_head, self.gauss_sd, self.headcount = True, 0.1, 1
        self.lamb, self.rank = 20, 0
        self.__dict__.update(kw)


def _digest(L):
    return hashlib.sha256(np.ascontiguousarray(L.astype(np.int32)).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["sk_ave_uniform", "sk_ave_peaked", "sk_k309_small", "sk_k400_ragged",
                                  "sk_gauss_per_head", "sk_vggsound_full"])
def test_optimize_L_sk_gpu_matches_reference_golden(golden_dir, name):
    from selavi_amd import sk_utils
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    N, K = int(g["N"]), int(g["K"])
    PS = synth_PS(N, K, float(g["scale"]), int(g["seed"]))
    args, head = Args(), 0
    if "dist_in" in g.files:
        head = int(g["head"])
        args = Args(distribution='gauss', headcount=g["dist_in"].shape[0],
                    dist=[torch.from_numpy(d.copy()).reshape(K, 1).cuda() for d in g["dist_in"]])
    P = torch.from_numpy(PS).cuda()
    cost, L = sk_utils.optimize_L_sk_gpu(args, P, head, None)
    info = sk_utils.optimize_L_sk_gpu.last_info
    L = L.cpu().numpy()
    assert info["iters"] == int(g["iters"])
    assert _digest(L) == bytes(g["digest"]).decode(), f"{(L[:4096] != g['labels_head']).sum()} of first 4096 differ"
    assert np.array_equal(np.bincount(L, minlength=K), g["hist"])
    assert abs(cost - float(g["cost"])) <= 1e-9 * abs(float(g["cost"]))
    np.testing.assert_allclose(info["alpha"].cpu().numpy(), g["alpha"], rtol=1e-9)
    if "dist_after" in g.files:      # args.dist mutated in place exactly like sk_utils.py:388
        np.testing.assert_array_equal(args.dist[head].cpu().numpy().ravel(), g["dist_after"][head])


@pytest.mark.parametrize("N,K", [(1, 5), (7, 1), (65, 64), (513, 65), (1000, 309), (300, 512)])
def test_edge_shapes_match_oracle(N, K):
    from selavi_amd import sk_utils
    PS = synth_PS(N, K, 1.5, 77 + N)
    cost_o, L_o, info_o = sk_ref.optimize_L_sk(PS)
    cost, L = sk_utils.optimize_L_sk_gpu(Args(), torch.from_numpy(PS).cuda(), 0, None)
    info = sk_utils.optimize_L_sk_gpu.last_info
    assert info["iters"] == info_o["iters"]
    assert np.array_equal(L.cpu().numpy(), L_o)
    assert abs(cost - cost_o) <= 1e-9 * max(1.0, abs(cost_o))


def test_max_iter_cap_and_determinism():
    from selavi_amd import sk_utils
    PS = synth_PS(2048, 309, 4.0, 5)
    P = torch.from_numpy(PS).cuda()
    r = torch.full((309,), 1.0 / 309, dtype=torch.float64, device="cuda")
    L1, ls1, i1 = sk_utils.sinkhorn(P.clone(), r, 20, max_iter=25)
    L2, ls2, i2 = sk_utils.sinkhorn(P.clone(), r, 20, max_iter=25)
    assert i1["iters"] == 25 and i2["iters"] == 25
    assert torch.equal(L1, L2) and torch.equal(i1["alpha"], i2["alpha"])      # fixed-order reductions
    _, L_o, info_o = sk_ref.optimize_L_sk(PS, max_iter=25)
    assert np.array_equal(L1.cpu().numpy(), L_o)


def test_prepare_fused_softmax_product():
    from selavi_amd import sk_utils
    for N, K in [(3328, 28), (1037, 309), (200, 400)]:
        lv, la = synth_logits(N, K, 3.0, 9)
        want = sk_ref.head_probabilities(lv, la)
        got = sk_utils.head_probabilities(torch.from_numpy(lv).cuda(), torch.from_numpy(la).cuda())
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-13, atol=0)
        got10 = sk_utils.head_probabilities(torch.from_numpy(lv).cuda(), torch.from_numpy(la).cuda(), power=10.0)
        np.testing.assert_allclose(got10.cpu().numpy(), want ** 10.0, rtol=2e-12, atol=0)
        sm = sk_utils.softmax64(torch.from_numpy(lv).cuda()).cpu().numpy()
        np.testing.assert_allclose(sm, sk_ref.softmax64(lv), rtol=1e-13)


def test_roundtrip_properties_full_size():
    """Size-independent properties at BASELINE's VGG-Sound size: after convergence the scaled
    matrix has row sums 1/N and column sums ~r (doubly-stochastic up to the 0.1 L1 tolerance)."""
    from selavi_amd import sk_utils
    N, K = 170752, 309
    g = torch.Generator(device="cuda").manual_seed(3)
    lv = torch.randn(N, K, device="cuda", generator=g) * 2
    la = torch.randn(N, K, device="cuda", generator=g) * 2
    P = sk_utils.head_probabilities(lv, la, power=10.0)
    r = torch.full((K,), 1.0 / K, dtype=torch.float64, device="cuda")
    L, logsum, info = sk_utils.sinkhorn(P, r, 20, already_powered=True)
    assert info["iters"] % 10 == 1 and info["iters"] < 2000
    Q = P * info["beta"][:, None] * info["alpha"][None, :]
    assert torch.allclose(Q.sum(1), torch.full((N,), 1.0 / N, dtype=torch.float64, device="cuda"), rtol=1e-9)
    assert (Q.sum(0) - r).abs().sum().item() < 0.11
    assert torch.equal(L, torch.argmax(Q, 1))
    assert L.min() >= 0 and L.max() < K
