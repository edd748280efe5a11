"""bf16 eval-mode trunk forward (selavi_amd/infer16.py, experimental opt-in): features against the fp32 eval path of
the same model.  bf16 activations through ~20 layers: a few 1e-3 of relative error, NOT bit-exact by design."""
import numpy as np
import pytest
import torch

from oracle.model_ref import portable_fill_, portable_init_

pytestmark = pytest.mark.gpu


def _model(hc=2, K=12):
    from selavi_amd import model as smodel
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    return m.cuda()


@pytest.mark.parametrize("shape", [(6, 4, 32, 40, 36), (3, 8, 112, 129, 100)])
def test_bf16_features_track_the_fp32_eval_forward(shape):
    from selavi_amd import infer16
    B, T, S, F, Tp = shape
    m = _model()
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, F, Tp), 6).cuda()
    m.train()
    with torch.no_grad():                       # realistic running statistics
        for _ in range(3):
            m(video, audio)
    m.eval()
    m.return_features = True
    with torch.no_grad():
        fv, fa = m(video, audio)
    gv, ga = infer16.Engine(m).features(video, audio)
    for name, got, want in (("video", gv, fv), ("audio", ga, fa)):
        assert got.shape == want.shape == (B, 512) and got.dtype == torch.float32
        rel = ((got - want).norm() / want.norm()).item()
        cos = torch.nn.functional.cosine_similarity(got, want, dim=1).min().item()
        print(f"{name}: relative L2 error {rel:.2e}, worst per-clip cosine {cos:.6f}")
        assert rel <= 3e-2 and cos >= 0.999, (name, rel, cos)


def test_bf16_feature_pass_gives_the_same_pseudo_labels_on_separable_data():
    """Joint argmax over the heads' logits from bf16 features vs fp32 features on a batch with class structure: the
    assignments agree except possibly for samples at a decision boundary."""
    from selavi_amd import clustering_metrics as cm, infer16
    m = _model(hc=1, K=8)
    g = torch.Generator().manual_seed(0)
    lab = torch.arange(32) % 8
    video = (torch.randn(32, 3, 4, 32, 32, generator=g) * 0.3 + (lab.view(-1, 1, 1, 1, 1) - 3.5) * 0.4).cuda()
    audio = (torch.randn(32, 1, 40, 36, generator=g) * 0.3 + (lab.view(-1, 1, 1, 1) - 3.5) * 0.4).cuda()
    m.train()
    with torch.no_grad():
        for _ in range(3):
            m(video, audio)
    m.eval()
    with torch.no_grad():
        lv, la = m(video, audio)
        gv, ga = infer16.Engine(m).features(video, audio)
        lv16, la16 = m.mlp_v.forward(gv), m.mlp_a.forward(ga)
    a, b = cm.joint_argmax(lv, la).cpu(), cm.joint_argmax(lv16, la16).cpu()
    agree = (a == b).float().mean().item()
    print(f"pseudo-label agreement bf16 vs fp32 features: {agree:.3f}")
    assert agree >= 0.9


@pytest.mark.parametrize("hc", [1, 2])
def test_cluster_round_with_bf16_feature_pass(hc):
    """sk_utils.cluster with args.feature_pass = "bf16": the Sinkhorn-Knopp round runs on bf16 features; its pseudo
    labels agree with the fp32 round's on (nearly) every sample of a synthetic dataset with class structure."""
    import argparse
    from selavi_amd import sk_utils
    from selavi_amd.data import SyntheticAVDataset
    from selavi_amd.utils import warmup_batchnorm
    ds = SyntheticAVDataset(n=128, T=4, S=32, F=40, Tp=36, n_classes=6)
    m = _model(hc=hc, K=6).train()
    loader = [(torch.stack([ds[i][0] for i in range(b, b + 16)]), torch.stack([ds[i][1] for i in range(b, b + 16)]))
              for b in range(0, 64, 16)]

    def args(fp):
        return argparse.Namespace(distribution="default", dist=None, diff_dist_every=False, diff_dist_per_head=True,
                                  gauss_sd=0.1, lamb=20, ind_groups=1, match=False, shuffle_sk_pass=False, headcount=hc,
                                  rank=0, world_size=1, feature_pass=fp)
    warmup_batchnorm(args(None), m, loader, batches=4)
    old = torch.zeros(128, hc, dtype=torch.long, device="cuda")
    np.random.seed(1)
    l32 = sk_utils.cluster(args(None), old, ds, m, 0, None, None, None, 0)
    np.random.seed(1)
    l16 = sk_utils.cluster(args("bf16"), old, ds, m, 0, None, None, None, 0)
    agree = (l32 == l16).float().mean().item()
    print(f"hc={hc}: SK pseudo-label agreement bf16 vs fp32 feature pass: {agree:.3f}")
    assert l16.shape == (128, hc) and agree >= 0.9
    assert m.training


def test_eval_dump_with_bf16_feature_pass(tmp_path):
    """get_clusters.get_cluster_assignments_gpu honours args.feature_pass too: head logits from bf16 features track the
    fp32 dump."""
    import argparse
    from selavi_amd import get_clusters
    from selavi_amd.data import SyntheticAVDataset
    ds = SyntheticAVDataset(n=16, T=4, S=32, F=40, Tp=36, n_classes=6)
    m = _model(hc=2, K=6).train()
    with torch.no_grad():
        for _ in range(2):
            m(torch.stack([ds[i][0] for i in range(8)]).cuda(), torch.stack([ds[i][1] for i in range(8)]).cuda())

    def dump(fp):
        a = argparse.Namespace(world_size=1, rank=0, batch_size=8, workers=0, headcount=2, output_dir=None, exp_desc="x",
                               feature_pass=fp)
        return get_clusters.get_cluster_assignments_gpu(a, ds, m)
    p32, p16 = dump(None), dump("bf16")
    for h in range(2):
        rel = ((p16[0][h] - p32[0][h]).norm() / p32[0][h].norm()).item()
        assert rel <= 3e-2, rel
    assert torch.equal(p16[1], p32[1])
