"""CPU tests: the oracle (oracle/) against the golden vectors produced by the executed reference
(tests/golden/make_golden.py) and against the documented torchvision invariants."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import model_ref, sk_ref, step_ref
from oracle.model_ref import portable_fill_, portable_init_
from tests._synth import synth_PS

SK_CASES = ["sk_ave_uniform", "sk_ave_peaked", "sk_k309_small", "sk_k400_ragged", "sk_gauss_per_head",
            "sk_vggsound_full"]


def _digest(L):
    return hashlib.sha256(np.ascontiguousarray(L.astype(np.int32)).tobytes()).hexdigest()


@pytest.mark.parametrize("name", SK_CASES)
def test_sk_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    N, K = int(g["N"]), int(g["K"])
    PS = synth_PS(N, K, float(g["scale"]), int(g["seed"]))
    kd = None
    if "dist_in" in g.files:
        head = int(g["head"])
        kd = sk_ref.marginals(K, N, PS, 'gauss', g["dist_in"][head])
        assert np.array_equal(kd, g["dist_after"][head])          # the in-place mutation of args.dist
    cost, L, info = sk_ref.optimize_L_sk(PS, lamb=int(g["lamb"]), K_dist=kd)
    assert info["iters"] == int(g["iters"])
    assert info["iters"] % 10 == 1                                # err only refreshed at 0,10,20,...
    assert _digest(L) == bytes(g["digest"]).decode()              # bit-exact labels
    assert np.array_equal(L[:4096], g["labels_head"]) and np.array_equal(L[-4096:], g["labels_tail"])
    assert np.array_equal(np.bincount(L, minlength=K), g["hist"])
    assert abs(cost - float(g["cost"])) <= 1e-12 * abs(float(g["cost"]))
    np.testing.assert_allclose(info["alpha"], g["alpha"], rtol=1e-12)


def test_sk_sharded_emulation_is_label_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "sk_ave_peaked.npz"))
    PS = synth_PS(int(g["N"]), int(g["K"]), float(g["scale"]), int(g["seed"]))
    for world in (2, 8):
        L, info = sk_ref.optimize_L_sk_sharded(PS, world)
        assert info["iters"] == int(g["iters"])
        assert np.array_equal(L, g["labels"])


def test_get_loss_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "get_loss.npz"))
    acts = [torch.from_numpy(a) for a in g["acts"]]
    tg = torch.from_numpy(g["targets"])
    assert abs(model_ref.get_loss(acts, tg, headcount=3).item() - float(g["loss_hc3"])) < 1e-6
    assert abs(model_ref.get_loss(acts[0], tg[:, 0], headcount=1).item() - float(g["loss_hc1"])) < 1e-6


def test_match_order_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "match_order.npz"))
    # the reference ran 2 restarts; replay its recorded pair stream restart by restart
    pairs = [tuple(p) for p in g["pairs"]]
    best_perm, best_cost, pos = np.arange(g["emb1"].shape[1]), np.abs(g["emb1"] - g["emb2"]).sum(), 0
    for _ in range(2):
        # count how many pairs this restart consumed: replay until the patience break
        perm, cost, used = _replay(g["emb1"], g["emb2"], pairs[pos:])
        pos += used
        if cost < best_cost:
            best_cost, best_perm = cost, perm
    assert pos == len(pairs)
    np.testing.assert_array_equal(g["w_after"], g["w_before"][best_perm])
    np.testing.assert_array_equal(g["b_after"], g["b_before"][best_perm])


def _replay(e1, e2, pairs, steps=3000, patience=1000):
    emb2 = np.array(e2)
    K = e1.shape[1]
    perm, last, used = np.arange(K), 0, 0
    for it in range(min(steps, len(pairs))):
        i, j = pairs[it]
        used += 1
        cur = np.abs(e1[:, i] - emb2[:, i]).sum() + np.abs(e1[:, j] - emb2[:, j]).sum()
        fut = np.abs(e1[:, i] - emb2[:, j]).sum() + np.abs(e1[:, j] - emb2[:, i]).sum()
        if cur - fut > 0:
            emb2[:, [i, j]] = emb2[:, [j, i]]
            perm[i], perm[j] = perm[j], perm[i]
            last = it
        if it - last > patience:
            break
    return perm, np.abs(e1 - np.asarray(e2)[:, perm]).sum(), used


def test_trunk_param_counts_and_keys(golden_dir):
    # torchvision's documented sizes: r2plus1d_18 (fc-400) 31 505 325; resnet18 11 689 512
    assert sum(p.numel() for p in model_ref.r2plus1d_18().parameters()) == 31505325
    assert sum(p.numel() for p in model_ref.resnet18().parameters()) == 11689512
    for hc, K, n in [(1, 28, 36754709), (10, 309, 44633345)]:
        m = model_ref.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        assert sum(p.numel() for p in m.parameters()) == n
        want = [l.split(" ")[0] for l in open(os.path.join(golden_dir, f"state_dict_keys_hc{hc}.txt"))]
        assert list(m.state_dict().keys()) == want


@pytest.mark.parametrize("fx", ["model_hc1_k28_mlp1", "model_hc3_k12_mlp1", "model_hc2_k7_mlp0"])
def test_model_oracle_matches_reference_golden(golden_dir, fx):
    g = np.load(os.path.join(golden_dir, fx + ".npz"))
    hc, K, use_mlp = int(g["hc"]), int(g["K"]), bool(g["use_mlp"])
    B, T, S = int(g["B"]), int(g["T"]), int(g["S"])
    torch.manual_seed(1)
    m = model_ref.load_model(use_mlp=use_mlp, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    assert len(m.state_dict()) == int(g["n_keys"])
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5)
    audio = portable_fill_(torch.empty(B, 1, 40, 36), 6)
    m.eval()
    with torch.no_grad():
        fv, fa = m(video, audio)
    fv = torch.stack(fv if hc > 1 else [fv]).numpy()
    np.testing.assert_allclose(fv, g["eval_v"], rtol=1e-4, atol=1e-5)
    m.train()
    opt = step_ref.make_optimizer(m)
    losses = []
    for _ in range(2):
        loss, _, _ = step_ref.train_step(m, opt, video, audio, torch.from_numpy(g["selflabels"]),
                                         torch.from_numpy(g["selected"]), hc)
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("post/"):
            np.testing.assert_allclose(sd[k[5:]].flatten()[:64].numpy(), g[k], rtol=1e-3, atol=1e-5)


def test_philox_oracle_known_answers():
    """oracle/philox_ref.py against the known-answer vectors Random123 publishes for philox4x32-10 (kat_vectors)."""
    from oracle.philox_ref import dropout_mask, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in philox4x32_10([ctr], key)[0]) == want
    m = dropout_mask(31, 1, 0.3, 200000)
    assert abs(m.mean() - 0.7) < 5e-3 and set(np.unique(m)) == {0.0, 1.0}
    assert not np.array_equal(m, dropout_mask(31, 2, 0.3, 200000))          # another offset, another mask
    np.testing.assert_array_equal(dropout_mask(31, 1, 0.3, 1000, first=777), m[777:1777])
