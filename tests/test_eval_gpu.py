"""Evaluation dumps: selavi_amd.clustering_metrics / get_clusters against the numbers the
reference's clustering_metrics.k_means printed on the same seeded logits (tests/golden/make_eval_golden.py)."""
import os
import pickle

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_metrics.npz")


def _synth(seed, N, K, heads, n_classes):                 # identical to tests/golden/make_eval_golden.py:synth
    g = np.random.RandomState(seed)
    labels = g.randint(0, n_classes, size=N)
    proto_v, proto_a = g.randn(n_classes, K), g.randn(n_classes, K)
    v = [(1.5 * proto_v[labels] + g.randn(N, K)).astype(np.float32) for _ in range(heads)]
    a = [(1.5 * proto_a[labels] + g.randn(N, K)).astype(np.float32) for _ in range(heads)]
    return v, labels * 3 + 1, a


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(4097, 309), (1000, 28), (513, 400), (1, 5)])
def test_joint_argmax_is_the_float64_softmax_product_argmax(N, K):
    from selavi_amd import clustering_metrics as cm
    g = torch.Generator().manual_seed(N + K)
    lv, la = torch.randn(N, K, generator=g) * 3, torch.randn(N, K, generator=g) * 3
    lv[0, :] = 0.0                                          # a row of exact ties -> first index
    la[0, :] = 0.0
    ref = (torch.softmax(lv, 1, dtype=torch.float64) * torch.softmax(la, 1, dtype=torch.float64)).argmax(1)
    got = cm.joint_argmax(lv.cuda(), la.cuda()).cpu()
    assert torch.equal(got, ref) and int(got[0]) == 0


@pytest.mark.gpu
def test_contingency_table_and_range_check():
    from selavi_amd import clustering_metrics as cm
    g = np.random.RandomState(0)
    p, t = g.randint(0, 309, size=200001), g.randint(0, 309, size=200001)
    got = cm.contingency(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), 309, 309).cpu().numpy()
    ref = np.zeros((309, 309), dtype=np.int64)
    np.add.at(ref, (p, t), 1)
    assert np.array_equal(got, ref) and got.sum() == 200001
    with pytest.raises(ValueError):
        cm.contingency(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), 309, 300)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
def test_k_means_metrics_match_the_reference_printout(name, tmp_path):
    from selavi_amd import clustering_metrics as cm
    d = np.load(GOLD)
    seed, N, K, heads, ncls = [int(x) for x in d[name + "_cfg"]]
    v, labels, a = _synth(seed, N, K, heads, ncls)
    if heads > 1:
        PS = [[torch.from_numpy(x) for x in v], torch.from_numpy(labels), [torch.from_numpy(x) for x in a]]
    else:
        PS = [torch.from_numpy(v[0]), torch.from_numpy(labels), torch.from_numpy(a[0])]
    path = tmp_path / "ps.pkl"
    pickle.dump(PS, open(path, "wb"))
    r = cm.k_means(path=str(path), ncentroids=K, use_all_heads=heads > 1, verbose=False)
    got = np.array([r["nmi"], r["anmi"], r["ari"], r["entropy"], r["purity"], r["acc"]])
    assert np.abs(got - d[name + "_metrics"]).max() < 1e-12, (got, d[name + "_metrics"])
    if heads > 1:
        assert np.abs(np.array(r["nmi_per_head"]) - d[name + "_head_nmi"]).max() < 1e-12


@pytest.mark.gpu
def test_get_clusters_dump_feeds_k_means(tmp_path):
    import argparse
    from selavi_amd import clustering_metrics as cm, get_clusters, model as smodel
    from selavi_amd.data import SyntheticAVDataset
    torch.manual_seed(0)
    ds = SyntheticAVDataset(n=24, T=4, S=32, F=40, Tp=50, n_classes=6)
    for hc in (1, 2):
        m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=6,
                              pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).cuda()
        args = argparse.Namespace(world_size=1, rank=0, batch_size=8, workers=0, headcount=hc,
                                  output_dir=str(tmp_path), exp_desc=f"dump{hc}")
        out = get_clusters.get_cluster_assignments_gpu(args, ds, m)
        PS = pickle.load(open(tmp_path / f"dump{hc}.pkl", "rb"))
        assert torch.equal(PS[1], torch.tensor(ds._labels)[:24].long())
        if hc == 1:
            assert PS[0].shape == (24, 6) and PS[0].dtype == torch.float64
        else:
            assert len(PS[0]) == 2 and PS[0][1].shape == (24, 6) and m.return_features is False
            # rank 0 applies the heads to the feature bank in eval mode: same logits as a plain eval forward
            m.eval()
            with torch.no_grad():
                v, a = m(torch.stack([ds[i][0] for i in range(8)]).cuda(), torch.stack([ds[i][1] for i in range(8)]).cuda())
            assert torch.allclose(PS[0][1][:8].float().cpu(), v[1].cpu(), atol=1e-4)
        r = cm.k_means(PS=PS, ncentroids=6, use_all_heads=hc > 1, verbose=False)
        assert 0.0 <= r["acc"] <= 1.0 and np.isfinite(r["nmi"])
