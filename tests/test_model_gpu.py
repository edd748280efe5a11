"""GPU parity tests of the full model / training step: HIP engine vs the CPU oracle and the golden
vectors of the executed reference.  Tolerance 1e-3 (BASELINE.json north star)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref, step_ref
from oracle.model_ref import portable_fill_, portable_init_

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["heuristic", "benchmark"])
def launch_configs(request):
    """The reference-generated goldens under BOTH ways a launch configuration is chosen: the built-in heuristics (tests,
    smoke, library users) and benchmark mode -- every candidate tile / K-split timed once per layer shape, the counterpart of
    the reference's cudnn.benchmark = True (main.py:187) and what bench.py's headline runs on."""
    from selavi_amd import ops
    prev = ops.benchmark
    ops.benchmark = request.param == "benchmark"
    ops.ConvPlan._cache.clear()
    yield request.param
    if request.param == "benchmark":          # (shapes an earlier test timed come from the in-process cache)
        assert len(ops._tune_cache()) > 0 and ops.benchmark, "benchmark mode configured nothing"
    ops.benchmark = prev
    ops.ConvPlan._cache.clear()


def _build(hc, K, use_mlp):
    from selavi_amd import model as smodel
    m = smodel.load_model(use_mlp=use_mlp, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    return m.cuda()


def _check_all_grads(m, golden_dir, fname="grads_hc1_k28.npz"):
    """Every parameter gradient vs the fp64 run of the reference model (tests/golden/grads_*.npz).

    Train-mode BatchNorm ResNets at random init have chaotic gradients: perturbations of 1e-7 (one
    fp32 rounding, a different summation order) come back as 1e-3..1e-2 in the parameter gradients.
    Measured on the reference itself: its fp32 CPU run deviates from its fp64 run by e_cpu = up to
    1.2e-2 (tiny fixture) / median 3.4e-3, max 5.6e-3 (wide fixture), and every HIP conv in every
    launch configuration is within 1e-6 of fp64 (tools/split_err.py) while the end-to-end gradient moves
    by 6e-3 between two split-K choices.  Which tensor the amplified noise lands on is arbitrary, so
    each tensor is held to 3 x the reference's own worst fp32 deviation -- "as accurate as the
    reference's fp32 arithmetic" -- on the first 256 elements and on the norm.  Real backward bugs show
    up as O(0.1..1) here and at 1e-4 in the per-op tests (test_ops_gpu.py)."""
    g = np.load(os.path.join(golden_dir, fname))
    names = [str(n) for n in g["names"]]
    params = dict(m.named_parameters())
    assert set(names) == set(params.keys())
    tol = 3 * float(np.max(g["e_cpu"]))
    errs = []
    for i, name in enumerate(names):
        gr = params[name].grad.detach().double().cpu()
        n = min(256, gr.numel())
        ref_norm = float(g["norms"][i])
        assert abs(gr.norm().item() - ref_norm) <= tol * ref_norm + 1e-12, (name, gr.norm().item(), ref_norm)
        head = torch.from_numpy(g["heads"][i, :n])
        err = (gr.flatten()[:n] - head).norm().item()
        # the head error is measured against the WHOLE tensor's rms so tiny leading entries don't dominate
        scale = ref_norm * (n / gr.numel()) ** 0.5 + 1e-30
        assert err <= tol * scale + 1e-12, (name, err, scale, tol)
        errs.append(err / scale)
    # the bulk as well: measured medians are 1.4e-4 .. 1.0e-2 depending on the launch configuration
    # (the reference's own: 8e-5 tiny / 3.4e-3 wide fixture); the per-op accuracy that is NOT subject
    # to this amplification is pinned at 5e-6 against fp64 in test_ops_gpu.py
    assert np.median(errs) <= tol, (np.median(errs), tol)
    print(f"head error / tensor rms: worst {max(errs):.2e} median {np.median(errs):.2e} "
          f"(reference fp32 noise: worst {np.max(g['e_cpu']):.2e} median {np.median(g['e_cpu']):.2e})")


def test_all_parameter_gradients_wide_fixture(golden_dir):
    """Gradients of all 190 parameter tensors at B=6, T=8, 64x64 video / 96x96 audio (>= 54 elements
    per channel in every BatchNorm) against the reference's fp64 run, and the loss to 1e-5."""
    from selavi_amd.utils import get_loss
    g = np.load(os.path.join(golden_dir, "grads_hc1_k28_wide.npz"))
    hc, K = int(g["hc"]), int(g["K"])
    B, T, S, FA, TA = (int(g[k]) for k in ("B", "T", "S", "FA", "TA"))
    m = _build(hc, K, True).train()
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, FA, TA), 6).cuda()
    selflabels = torch.from_numpy(g["selflabels"]).cuda()
    selected = torch.from_numpy(g["selected"]).cuda()
    fv, fa = m(video, audio)
    labels = selflabels[selected, 0]
    loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["loss64"]), rtol=1e-5)
    _check_all_grads(m, golden_dir, "grads_hc1_k28_wide.npz")


def _stack(x):
    return torch.stack(list(x)) if isinstance(x, (list, tuple)) else x.unsqueeze(0)


def test_state_dict_keys_match_reference(golden_dir):
    from selavi_amd import model as smodel
    for hc, K in [(1, 28), (10, 309)]:
        m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        lines = [l.split(" ", 1) for l in open(os.path.join(golden_dir, f"state_dict_keys_hc{hc}.txt"))]
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _ in lines]
        for (k, shp), v in zip(lines, sd.values()):
            assert list(v.shape) == eval(shp), k


@pytest.mark.parametrize("fx", ["model_hc1_k28_mlp1", "model_hc3_k12_mlp1", "model_hc2_k7_mlp0"])
def test_model_and_two_train_steps_match_reference_golden(golden_dir, fx, launch_configs):
    from selavi_amd import optim, train
    g = np.load(os.path.join(golden_dir, fx + ".npz"))
    hc, K, use_mlp = int(g["hc"]), int(g["K"]), bool(g["use_mlp"])
    B, T, S = int(g["B"]), int(g["T"]), int(g["S"])
    m = _build(hc, K, use_mlp)
    assert len(m.state_dict()) == int(g["n_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, 40, 36), 6).cuda()
    m.eval()
    with torch.no_grad():
        fv, fa = m(video, audio)
        m.return_features = True
        gv, ga = m(video, audio)
        m.return_features = False
    np.testing.assert_allclose(_stack(fv).cpu().numpy(), g["eval_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(_stack(fa).cpu().numpy(), g["eval_a"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(gv.cpu().numpy(), g["feat_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ga.cpu().numpy(), g["feat_a"], rtol=1e-3, atol=1e-3)
    # two training steps (main.py:284-302 with the fused SGD)
    m.train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    selflabels = torch.from_numpy(g["selflabels"]).cuda()
    selected = torch.from_numpy(g["selected"]).cuda()
    losses = []
    for step in range(2):
        if step == 0:
            fv, fa = m(video, audio)
            np.testing.assert_allclose(_stack(fv).detach().cpu().numpy(), g["train_v"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(_stack(fa).detach().cpu().numpy(), g["train_a"], rtol=1e-3, atol=1e-3)
            # undo the running-stat side effect of this probe forward: reload buffers
            m2 = _build(hc, K, use_mlp)
            m.load_state_dict(m2.state_dict())
            for mod in m.modules():
                if hasattr(mod, "_pending"):
                    mod._pending = 0
            if True:
                # gradient check on the first step
                from selavi_amd.utils import get_loss
                fv, fa = m(video, audio)
                labels = selflabels[selected, 0] if hc == 1 else selflabels[selected, :]
                loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
                opt.zero_grad()
                loss.backward()
                if fx == "model_hc1_k28_mlp1":
                    _check_all_grads(m, golden_dir)
                w_before = {k: v.detach().clone() for k, v in m.named_parameters()}
                g_before = {k: v.grad.detach().clone() for k, v in m.named_parameters()}
                opt.step()
                for k, v in m.named_parameters():     # first SGD step: p -= lr * (g + wd * p)
                    want = w_before[k] - 1e-2 * (g_before[k] + 1e-5 * w_before[k])
                    assert torch.allclose(v.detach(), want, rtol=1e-5, atol=1e-6), k
                losses.append(loss.item())
        else:
            # the oracle evaluated AT THE HIP WEIGHTS after step 1 (train-mode forward + loss on CPU):
            # isolates the step-2 forward/loss parity from the chaotic gradient (see below)
            o = model_ref.load_model(use_mlp=use_mlp, num_classes=K, norm_feat=False, headcount=hc)
            step_ref.set_dropout_p(o, 0.0)
            o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
            o.train()
            with torch.no_grad():
                ov, oa = o(video.cpu(), audio.cpu())
                lab = (selflabels[selected, 0] if hc == 1 else selflabels[selected, :]).cpu()
                loss2_oracle = float(0.5 * model_ref.get_loss(ov, lab, hc) + 0.5 * model_ref.get_loss(oa, lab, hc))
            losses.append(train.train_step(m, opt, video, audio, selflabels, selected, hc).item())
    # Step 1 is a pure forward: 1e-3 (measured ~3e-6).  Step 2 is checked exactly where it can be: against
    # the oracle's loss at the SAME (HIP) weights (1e-3), plus the SGD update itself above.  Against the
    # golden step-2 loss only a band is meaningful: the gradient of this tiny fixture (16 elements per
    # channel in layer4's BatchNorms) is reproducible to ~1e-2 between fp32 implementations (see
    # _check_all_grads), the step changes the loss by 0.9, and HIP builds that differ only in summation
    # order (K-split choice, tap-major K) landed at 2.636, 2.659 and 2.709 (reference fp32 2.629, fp64 2.635).
    np.testing.assert_allclose(losses[0], g["losses"][0], rtol=1e-3)
    np.testing.assert_allclose(losses[1], loss2_oracle, rtol=1e-3)
    np.testing.assert_allclose(losses[1], g["losses"][1], rtol=6e-2)
    # Weights after TWO steps are chaotic at this lr (the step-2 gradient norm of the stem is 186 in the
    # reference's fp64 run and 154 in its fp32 run), so only the BN running statistics are compared
    # after step 2; the SGD update itself is checked after step 1 in _check_all_grads / below.
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("post/") and "running_" in k:
            np.testing.assert_allclose(sd[k[5:]].flatten()[:64].cpu().numpy(), g[k], rtol=5e-2, atol=5e-3)
    assert int(sd["video_network.base.stem.1.num_batches_tracked"]) == 2


def test_dropout_masks_and_larger_batch_match_oracle():
    """Train-mode logits with an injected dropout mask (p = 0.3) at B = 5, T = 6, hc = 2."""
    hc, K, B = 2, 9, 5
    m = _build(hc, K, True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.3
    o = model_ref.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(o, seed=31)
    video = portable_fill_(torch.empty(B, 3, 6, 24, 24), 15)
    audio = portable_fill_(torch.empty(B, 1, 33, 30), 16)
    g = torch.Generator().manual_seed(4)
    m1 = torch.bernoulli(torch.full((2 * hc, B, 512), 0.7), generator=g)
    m2 = torch.bernoulli(torch.full((2 * hc, B, 512), 0.7), generator=g)
    m._dropout_masks = (m1.cuda(), m2.cuda())
    m.train()
    fv, fa = m(video.cuda(), audio.cuda())
    # oracle with the same masks: run trunks, then heads by hand
    o.train()
    o.return_features = True
    with torch.no_grad():
        gv, ga = o(video, audio)
    heads = [getattr(o, f"mlp_v{h}") for h in range(hc)] + [getattr(o, f"mlp_a{h}") for h in range(hc)]
    for gi, hd in enumerate(heads):
        x = (gv if gi < hc else ga) * m1[gi] / 0.7
        bf = hd.block_forward
        h1 = bf[2](x)
        a = torch.relu(torch.nn.functional.batch_norm(h1, None, None, bf[4].weight, bf[4].bias, True, 0.1, 1e-5))
        want = bf[8](a * m2[gi] / 0.7)
        got = (fv if gi < hc else fa)[gi % hc]
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-3, atol=2e-3)


def test_cfg4_shaped_odd_temporal_extent_matches_oracle():
    """cfg4's shape family at reduced size: T = 30 (-> 15 -> 8 -> 4 through the stride-2 temporal convs,
    odd extents and empty stride-parity classes), K = 400, hc = 2, 257-bin spectrogram width class.
    Eval-mode and train-mode (batch statistics) logits + features vs the oracle, and a state_dict
    round trip (main.py:227 / utils.py:247 resume path) into a second model."""
    from selavi_amd import model as smodel
    hc, K, B = 2, 400, 2
    m = _build(hc, K, True)
    o = model_ref.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(o, seed=31)
    step_ref.set_dropout_p(o, 0.0)
    torch.set_num_threads(min(8, os.cpu_count()))
    video = portable_fill_(torch.empty(B, 3, 30, 32, 32), 25)
    audio = portable_fill_(torch.empty(B, 1, 65, 50), 26)
    for mode in ("eval", "train"):
        getattr(m, mode)(), getattr(o, mode)()
        with torch.no_grad():
            fv, fa = m(video.cuda(), audio.cuda())
            wv, wa = o(video, audio)
        for got, want in zip(list(fv) + list(fa), list(wv) + list(wa)):
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-3, atol=2e-3)
    # running statistics moved identically (one train-mode forward each)
    sd_m, sd_o = m.state_dict(), o.state_dict()
    for k in ("video_network.base.layer3.0.conv1.0.1.running_var", "audio_network.base.layer4.0.bn2.running_mean",
              "video_network.base.stem.4.running_mean"):
        np.testing.assert_allclose(sd_m[k].cpu().numpy(), sd_o[k].numpy(), rtol=2e-3, atol=1e-4)
    # checkpoint round trip: a fresh model loaded from the state_dict reproduces the eval output bit for bit
    m2 = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc).cuda()
    m2.load_state_dict({k: v.clone() for k, v in sd_m.items()})
    m.eval(), m2.eval()
    with torch.no_grad():
        a1, b1 = m(video.cuda(), audio.cuda())
        a2, b2 = m2(video.cuda(), audio.cuda())
    for x, y in zip(list(a1) + list(b1), list(a2) + list(b2)):
        assert torch.equal(x, y)


def test_checkpoint_resume_continues_bit_identically():
    """main.py:221-233 / utils.py:230-275 save and restore {model, optimizer} state_dicts: after
    step 1 -> save -> (fresh model + fresh optimizer).load_state_dict -> step 2 must equal the uninterrupted
    run bit for bit (momentum buffers under torch.optim.SGD's 'momentum_buffer' key)."""
    from selavi_amd import model as smodel, optim, train
    hc, K, B = 1, 7, 2
    video = portable_fill_(torch.empty(B, 3, 4, 32, 32), 35).cuda()
    audio = portable_fill_(torch.empty(B, 1, 40, 36), 36).cuda()
    selflabels = torch.arange(64).remainder(K).view(64, 1).cuda()
    selected = torch.tensor([5, 11]).cuda()
    m = _build(hc, K, True).train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    train.train_step(m, opt, video, audio, selflabels, selected, hc)
    ck = {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": opt.state_dict()}
    st = ck["opt"]["state"]
    assert len(st) == len(list(m.parameters())) and all("momentum_buffer" in v for v in st.values())
    ck["opt"] = {"state": {k: {"momentum_buffer": v["momentum_buffer"].clone()} for k, v in st.items()},
                 "param_groups": ck["opt"]["param_groups"]}
    l2 = train.train_step(m, opt, video, audio, selflabels, selected, hc)
    m2 = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc).cuda().train()
    step_ref.set_dropout_p(m2, 0.0)
    m2.load_state_dict(ck["model"])
    opt2 = optim.SGD(m2.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    opt2.load_state_dict(ck["opt"])
    l2b = train.train_step(m2, opt2, video, audio, selflabels, selected, hc)
    assert float(l2) == float(l2b)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_training_step_is_bit_reproducible_run_to_run():
    """Six repetitions of {two steps from the same state} (audio trunk on its own stream, dropout on)
    give bit-identical parameters, running statistics and losses.  Guards the fixed-order reductions
    and the kernels' LDS hand-offs: a missing barrier after the weight-gradient kernel's LDS parameter
    fill made ~1 run in 5 differ at the 1e-3 level before it was found with this check."""
    import hashlib
    from selavi_amd import optim, train
    hc, K, B = 2, 31, 4
    m = _build(hc, K, True).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.3
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    video = portable_fill_(torch.empty(B, 3, 8, 64, 64), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, 65, 50), 6).cuda()
    selflabels = (torch.arange(64 * hc).view(64, hc) * 7 % K).cuda()
    selected = torch.arange(B).cuda() * 3
    sigs = set()
    for _ in range(6):
        m.load_state_dict(state0)
        torch.manual_seed(0)
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        losses = [float(train.train_step(m, opt, video, audio, selflabels, selected, hc)) for _ in range(2)]
        h = hashlib.sha256(repr(losses).encode())
        for v in m.state_dict().values():
            h.update(v.detach().cpu().numpy().tobytes())
        sigs.add(h.hexdigest())
    assert len(sigs) == 1


def test_single_head_forward_on_feature_bank():
    """head.forward(N x 512 bank) in eval mode -- the call sk_utils.py:309-312 makes."""
    m = _build(3, 12, True)
    o = model_ref.load_model(use_mlp=True, num_classes=12, norm_feat=False, headcount=3)
    portable_init_(o, seed=31)
    m.eval(), o.eval()
    bank = portable_fill_(torch.empty(1000, 512), 3)
    with torch.no_grad():
        got = m.mlp_a1.forward(bank.cuda()).cpu()
        want = o.mlp_a1.forward(bank)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("hc,use_mlp,norm_feat", [(1, True, False), (3, True, True), (2, False, False)])
def test_single_clip_batch_keeps_the_reference_shapes(hc, use_mlp, norm_feat):
    """B == 1 (model.py:223-231): `.squeeze()` drops the batch dimension of the trunk features, `return_features`
    hands those 1-D tensors out as they are, the head path re-adds the dimension.  Eval mode, against the oracle."""
    from selavi_amd import model as smodel
    m = smodel.load_model(use_mlp=use_mlp, num_classes=12, norm_feat=norm_feat, headcount=hc)
    o = model_ref.load_model(use_mlp=use_mlp, num_classes=12, norm_feat=norm_feat, headcount=hc)
    portable_init_(m, seed=31), portable_init_(o, seed=31)
    m = m.cuda().eval()
    o.eval()
    video = portable_fill_(torch.empty(1, 3, 4, 32, 32), 5)
    audio = portable_fill_(torch.empty(1, 1, 40, 36), 6)
    with torch.no_grad():
        gv, ga = m(video.cuda(), audio.cuda())
        wv, wa = o(video, audio)
        got = [gv, ga] if hc == 1 else list(gv) + list(ga)
        want = [wv, wa] if hc == 1 else list(wv) + list(wa)
        for g_, w_ in zip(got, want):
            assert tuple(g_.shape) == tuple(w_.shape) == (1, 12)
            np.testing.assert_allclose(g_.cpu().numpy(), w_.numpy(), rtol=1e-3, atol=1e-3)
        m.return_features = o.return_features = True
        fv, fa = m(video.cuda(), audio.cuda())
        rv, ra = o(video, audio)
        assert tuple(fv.shape) == tuple(rv.shape) == (512,) and tuple(fa.shape) == tuple(ra.shape) == (512,)
        np.testing.assert_allclose(fv.cpu().numpy(), rv.numpy(), rtol=1e-3, atol=1e-3)


def test_full_cfg2_step_reproducible_and_learning():
    """BASELINE cfg2 size (B=16, 16x112x112 video, 129x100 log-mel, K=309, hc=10), where the CPU oracle
    is unaffordable: (1) two runs of 6 steps from the same state are bit-identical (benchmark-mode
    configurations pinned by reusing the plans), (2) the first loss is ln(K) to 10 % (random init,
    near-uniform softmax), (3) fitting one fixed batch lowers the loss, (4) no non-finite parameter."""
    import math
    from selavi_amd import model as smodel, ops, optim, train
    hc, K, B = 10, 309, 16
    torch.manual_seed(31)
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc).cuda().train()
    step_ref.set_dropout_p(m, 0.0)                    # (3) wants a monotone signal
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator(device="cuda").manual_seed(7)
    video = torch.randn(B, 3, 16, 112, 112, device="cuda", generator=g)
    audio = torch.randn(B, 1, 129, 100, device="cuda", generator=g)
    selflabels = torch.randint(0, K, (1024, hc), device="cuda", generator=g)
    selected = torch.arange(B, device="cuda") * 5
    runs = []
    for _ in range(2):
        m.load_state_dict(state0)
        torch.manual_seed(3)                          # dropout masks
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        losses = [float(train.train_step(m, opt, video, audio, selflabels, selected, hc)) for _ in range(6)]
        runs.append((losses, [v.clone() for v in m.state_dict().values()]))
    assert runs[0][0] == runs[1][0]
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    losses = runs[0][0]
    assert abs(losses[0] - math.log(K)) <= 0.10 * math.log(K), losses
    assert losses[-1] < losses[0] - 0.05, losses
    assert all(torch.isfinite(v).all() for v in runs[0][1] if v.is_floating_point())


def test_cfg1_full_size_matches_executed_reference(golden_dir):
    """BASELINE configs[0] at full size (bs=4, 8x112x112 clips, 1x40x100 log-mel, K=28, hc=1): logits,
    trunk features, loss and running statistics against tests/golden/cfg1_full.npz, which
    make_golden.py produced by executing the reference's model.py / utils.py on CPU.  1e-3, fp32."""
    from selavi_amd.utils import get_loss
    g = np.load(os.path.join(golden_dir, "cfg1_full.npz"))
    hc, K, B = int(g["hc"]), int(g["K"]), int(g["B"])
    m = _build(hc, K, True)
    video = portable_fill_(torch.empty(B, 3, 8, 112, 112), 45).cuda()
    audio = portable_fill_(torch.empty(B, 1, 40, 100), 46).cuda()
    m.eval()
    with torch.no_grad():
        fv, fa = m(video, audio)
        m.return_features = True
        gv, ga = m(video, audio)
        m.return_features = False
    for got, key in ((fv, "eval_v"), (fa, "eval_a"), (gv, "feat_v"), (ga, "feat_a")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-3, atol=1e-3)
    m.train()
    fv, fa = m(video, audio)
    np.testing.assert_allclose(fv.detach().cpu().numpy(), g["train_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fa.detach().cpu().numpy(), g["train_a"], rtol=1e-3, atol=1e-3)
    labels = torch.from_numpy(g["selflabels"]).cuda()[torch.from_numpy(g["selected"]).cuda(), 0]
    loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-4)
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("post/"):
            np.testing.assert_allclose(sd[k[5:]].cpu().numpy(), g[k], rtol=1e-3, atol=1e-5)


def test_cfg2_full_size_forward_matches_executed_reference(golden_dir, launch_configs):
    """BASELINE configs[1] at full size -- the configuration the headline metric is quoted on (bs=16,
    16x112x112 clips, 1x129x100 log-mel, K=309, hc=10): train-mode logits of heads 0 and 9 of both
    modalities, the loss of main.py:284-293 and the eval-mode trunk features after that one training-mode
    forward, against tests/golden/cfg2_full.npz from the executed reference.  1e-3, fp32."""
    from selavi_amd.utils import get_loss
    g = np.load(os.path.join(golden_dir, "cfg2_full.npz"))
    hc, K, B = int(g["hc"]), int(g["K"]), int(g["B"])
    m = _build(hc, K, True).train()
    video = portable_fill_(torch.empty(B, 3, 16, 112, 112), 55).cuda()
    audio = portable_fill_(torch.empty(B, 1, 129, 100), 56).cuda()
    with torch.no_grad():
        fv, fa = m(video, audio)
        labels = torch.from_numpy(g["selflabels"]).cuda()[torch.from_numpy(g["selected"]).cuda(), :]
        loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
    for got, key in ((fv[0], "train_v0"), (fv[9], "train_v9"), (fa[0], "train_a0"), (fa[9], "train_a9")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-4)
    m.eval()
    m.return_features = True
    with torch.no_grad():
        gv, ga = m(video, audio)
    np.testing.assert_allclose(gv.cpu().numpy(), g["feat_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ga.cpu().numpy(), g["feat_a"], rtol=1e-3, atol=1e-3)


def test_cfg4_full_size_forward_matches_executed_reference(golden_dir):
    """BASELINE configs[3] (Kinetics-400 shape) at full per-GPU size: bs=16, 30x112x112 clips (temporal sizes
    30/15/8/4: every stride-2 temporal conv sees an odd or non-power-of-two length), 1x129x100 log-mel, K=400, hc=10.
    Train-mode logits, loss and eval features against tests/golden/cfg4_full.npz from the executed reference."""
    from selavi_amd.utils import get_loss
    g = np.load(os.path.join(golden_dir, "cfg4_full.npz"))
    hc, K, B, T = int(g["hc"]), int(g["K"]), int(g["B"]), int(g["T"])
    assert (hc, K, B, T) == (10, 400, 16, 30)
    m = _build(hc, K, True).train()
    video = portable_fill_(torch.empty(B, 3, T, 112, 112), 55).cuda()
    audio = portable_fill_(torch.empty(B, 1, 129, 100), 56).cuda()
    with torch.no_grad():
        fv, fa = m(video, audio)
        labels = torch.from_numpy(g["selflabels"]).cuda()[torch.from_numpy(g["selected"]).cuda(), :]
        loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
    for got, key in ((fv[0], "train_v0"), (fv[9], "train_v9"), (fa[0], "train_a0"), (fa[9], "train_a9")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-4)
    m.eval()
    m.return_features = True
    with torch.no_grad():
        gv, ga = m(video, audio)
    np.testing.assert_allclose(gv.cpu().numpy(), g["feat_v"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ga.cpu().numpy(), g["feat_a"], rtol=1e-3, atol=1e-3)


def test_full_size_gradients_within_reference_noise(golden_dir, launch_configs):
    """Backward at the headline configuration's FULL input size (bs 16, 16x112x112 video, 1x129x100 log-mel;
    hc=1, K=28 heads): all 190 parameter gradients against the reference's fp64 run
    (tests/golden/grads_cfg2_full.npz, ~10 min of CPU in make_golden.py --only-cfg2-grads).  Even at this size
    the reference's own fp32 run deviates from fp64 by a median 7.8e-3 / max 1.1e-2 (train-mode BN ResNet at
    random init), so the bar is the same noise-derived one as for the small fixtures; the loss agrees to 1e-5."""
    from selavi_amd.utils import get_loss
    g = np.load(os.path.join(golden_dir, "grads_cfg2_full.npz"))
    hc, K = int(g["hc"]), int(g["K"])
    B, T, S, FA, TA = (int(g[k]) for k in ("B", "T", "S", "FA", "TA"))
    assert (B, T, S, FA, TA) == (16, 16, 112, 129, 100)
    m = _build(hc, K, True).train()
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, FA, TA), 6).cuda()
    selflabels = torch.from_numpy(g["selflabels"]).cuda()
    selected = torch.from_numpy(g["selected"]).cuda()
    fv, fa = m(video, audio)
    labels = selflabels[selected, 0]
    loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["loss64"]), rtol=1e-5)
    _check_all_grads(m, golden_dir, "grads_cfg2_full.npz")


@pytest.mark.gpu
def test_training_step_with_batch_sliced_convs(monkeypatch):
    """The whole step with every large conv forced into batch slices (the path BASELINE configs[4]'s 128 x 32-frame
    per-GPU batch takes) against the same step unsliced: logits to 1e-4, loss to 1e-5, the update stays finite and close."""
    from selavi_amd import model as smodel, ops, optim, train

    def run(limit):
        monkeypatch.setattr(ops, "CONV_BUF_LIMIT", limit)
        monkeypatch.setattr(ops.ConvPlan, "_cache", {})
        torch.manual_seed(5)
        m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=12,
                              pretrained=False, norm_feat=False, use_max_pool=False, headcount=2).cuda()
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        g = torch.Generator(device="cuda").manual_seed(3)
        video = torch.randn(6, 3, 8, 48, 48, device="cuda", generator=g)
        audio = torch.randn(6, 1, 40, 60, device="cuda", generator=g)
        labels = torch.randint(0, 12, (64, 2), device="cuda", generator=g)
        sel = torch.arange(6, device="cuda")
        m.train()
        fv, fa = m(video, audio)
        logits = torch.stack(fv + fa).detach().clone()
        loss = train.train_step(m, opt, video, audio, labels, sel, 2)
        sliced = sum(p.chunks is not None for p in ops.ConvPlan._cache.values())
        return logits, float(loss), torch.cat([p.detach().flatten() for p in m.parameters()]), sliced

    l0, loss0, w0, n0 = run(0xFFFFFFF0)
    l1, loss1, w1, n1 = run(3 * 1024 * 1024)                       # layer-1 tensors are 5-13 MB here: 2-5 slices
    assert n0 == 0 and n1 >= 8
    assert (l0 - l1).abs().max().item() <= 1e-4 * max(1.0, l0.abs().max().item())
    assert abs(loss0 - loss1) <= 1e-5 * max(1.0, abs(loss0))
    assert torch.isfinite(w1).all() and (w0 - w1).norm().item() <= 1e-3 * w0.norm().item()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_step_replays_bit_identically_to_eager(precision):
    """train.GraphedStep: the whole step (forward, loss, backward with the weight-gradient side streams AND the audio
    trunk on its own stream -- its autograd node forks and joins with events --, SGD) captured into one HIP graph; two
    replays leave exactly the weights two eager steps leave, on the fp32 path and with the video trunk on the 16-bit path."""
    from selavi_amd import model as smodel, optim, train

    def make():
        m = smodel.load_model(use_mlp=True, num_classes=12, norm_feat=False, headcount=2)
        portable_init_(m, seed=31)
        step_ref.set_dropout_p(m, 0.0)
        m = m.cuda().train()
        m.set_precision(precision)
        assert m.overlap_audio
        return m, optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)

    video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
    audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
    sl = torch.from_numpy((np.arange(64 * 2).reshape(64, 2) * 7919 % 12).astype(np.int64)).cuda()
    sel = torch.tensor([3, 17, 42, 63]).cuda()
    m0, o0 = make()
    for _ in range(5):
        l0 = train.train_step(m0, o0, video, audio, sl, sel, 2)
    m1, o1 = make()
    gs = train.GraphedStep(m1, o1, video, audio, sl, sel, 2, warmup=3)
    for _ in range(2):
        l1 = gs.replay()
    torch.cuda.synchronize()
    assert float(l0) == float(l1)
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.gpu
def test_graphed_step_draws_fresh_dropout_masks_on_every_replay():
    """Dropout(0.3) under train.GraphedStep: the Philox key / offset live in device memory (slv_dropout_masks_dev) and the
    captured launch advances the offset, so replays draw DIFFERENT masks (host scalars would be frozen into the graph).
    (a) the captured draw against the numpy Philox oracle at offset, offset + 1, offset + 2; (b) a captured step with
    lr = 0 (weights frozen, same batch): the loss still changes from replay to replay -- only the masks can do that."""
    from oracle.philox_ref import dropout_mask
    from selavi_amd import model as smodel, nn as snn, optim, train
    from selavi_amd._lib import C, ptr, stream
    state = torch.tensor([31, 5], dtype=torch.int64, device="cuda")
    m1, m2 = torch.empty(4099, device="cuda"), torch.empty(513, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        C.slv_dropout_masks_dev(ptr(state), 0.3, ptr(m1), m1.numel(), ptr(m2), m2.numel(), stream())
    for k in range(3):
        gr.replay()
        torch.cuda.synchronize()
        want = dropout_mask(31, 5 + k, 0.3, 4099 + 513)
        np.testing.assert_array_equal(m1.cpu().numpy(), want[:4099])
        np.testing.assert_array_equal(m2.cpu().numpy(), want[4099:])
    assert state.tolist() == [31, 8]

    torch.manual_seed(7)
    m = smodel.load_model(use_mlp=True, num_classes=12, norm_feat=False, headcount=2)
    portable_init_(m, seed=31)
    m = m.cuda().train()                            # Dropout(0.3) as the reference has it (model.py:79,85)
    opt = optim.SGD(m.parameters(), lr=0.0, momentum=0.9, weight_decay=0.0)
    video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
    audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
    sl = torch.from_numpy((np.arange(64 * 2).reshape(64, 2) * 7919 % 12).astype(np.int64)).cuda()
    sel = torch.tensor([3, 17, 42, 63]).cuda()
    gs = train.GraphedStep(m, opt, video, audio, sl, sel, 2, warmup=2)
    st0 = snn.dropout_device_state(video.device).clone()
    losses = []
    for _ in range(4):
        losses.append(float(gs.replay()))
    torch.cuda.synchronize()
    assert len(set(losses)) == 4, losses            # frozen masks would repeat one value
    assert (snn.dropout_device_state(video.device) - st0).tolist() == [0, 4]


@pytest.mark.gpu
def test_dropout_masks_are_philox_bit_exact_and_loss_total():
    """slv_dropout_masks against the numpy Philox4x32-10 oracle (bit-exact, both masks of one launch, ragged sizes),
    and the in-library loss mean against the per-row losses."""
    from oracle.philox_ref import dropout_mask
    from selavi_amd._lib import C, ptr, stream
    for seed, off, p, n1, n2 in [(31, 1, 0.3, 20 * 16 * 512, 20 * 16 * 512), (2 ** 40 + 7, 2 ** 33 + 5, 0.5, 1001, 37),
                                 (0, 0, 0.0, 5, 0)]:
        m1 = torch.empty(n1, device="cuda")
        m2 = torch.empty(max(n2, 1), device="cuda")
        C.slv_dropout_masks(seed, off, p, ptr(m1), n1, ptr(m2) if n2 else 0, n2, stream())
        want = dropout_mask(seed, off, p, n1 + n2)
        np.testing.assert_array_equal(m1.cpu().numpy(), want[:n1])
        if n2:
            np.testing.assert_array_equal(m2[:n2].cpu().numpy(), want[n1:])
    rows = torch.rand(321, device="cuda")
    tot = torch.empty((), device="cuda")
    C.slv_heads_ce_total(ptr(rows), 321, 1.0 / 321, ptr(tot), stream())
    assert abs(float(tot) - float(rows.double().mean())) < 1e-6
    # the model draws its masks from this generator, keyed from torch's default generator: two train-mode forwards
    # differ, torch.manual_seed replays them (opt.py:152 / utils.py:277-283 semantics)
    from selavi_amd import model as smodel
    m = smodel.load_model(use_mlp=True, num_classes=7, norm_feat=False, headcount=2).cuda().train()
    v, a = torch.randn(4, 3, 4, 32, 32, device="cuda"), torch.randn(4, 1, 40, 36, device="cuda")
    outs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        with torch.no_grad():
            o1 = m(v, a)[0].stacked.clone()
            o2 = m(v, a)[0].stacked.clone()
        assert not torch.equal(o1, o2)
        outs.append(o1)
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


@pytest.mark.gpu
@pytest.mark.parametrize("B,IN,OUT,shared", [(16, 512, 512, 1), (5, 512, 309, 1), (40, 512, 309, 0), (128, 512, 512, 1),
                                             (70, 64, 20, 0)])
def test_grouped_head_linears_on_the_matrix_cores(B, IN, OUT, shared):
    """slv_heads_linear_fwd / _bwd_w / _bwd_x (the MFMA kernels: IN % 16 == 0) against fp64 matmuls with the same dropout
    masks: ragged batch tiles, K = 309 outputs (not a multiple of 4 or 16), X shared per modality or one per head."""
    from selavi_amd import ops
    from selavi_amd._lib import C, ptr, stream
    hc, G = 3, 6
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + OUT)
    X = torch.randn(2 if shared else G, B, IN, generator=g)
    Ws = [torch.randn(OUT, IN, generator=g) * IN ** -0.5 for _ in range(G)]
    bs = [torch.randn(OUT, generator=g) for _ in range(G)]
    mask = (torch.rand(G, B, IN, generator=g) > 0.3).float()
    msc = 1.0 / 0.7
    dout = torch.randn(G, B, OUT, generator=g)
    Xd, Wd, bd, md, dd = X.to(dev), [w.to(dev) for w in Ws], [b.to(dev) for b in bs], mask.to(dev), dout.to(dev)
    Wt, bt = ops.PtrArray(Wd), ops.PtrArray(bd)
    for use_mask in (True, False):
        Xm = torch.stack([X[gi // hc if shared else gi].double() * (mask[gi].double() * msc if use_mask else 1.0) for gi in range(G)])
        want = torch.stack([Xm[gi] @ Ws[gi].double().t() + bs[gi].double() for gi in range(G)])
        out = torch.empty(G, B, OUT, device=dev)
        C.slv_heads_linear_fwd(ptr(Xd), shared, hc, ptr(md) if use_mask else 0, msc, Wt.p, bt.p, ptr(out), G, B, IN, OUT, stream())
        assert float((out.cpu().double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        dW, db = torch.empty(G, OUT, IN, device=dev), torch.empty(G, OUT, device=dev)
        C.slv_heads_linear_bwd_w(ptr(dd), ptr(Xd), shared, hc, ptr(md) if use_mask else 0, msc, ptr(dW), ptr(db), G, B, IN, OUT, stream())
        wantW = torch.stack([dout[gi].double().t() @ Xm[gi] for gi in range(G)])
        assert float((dW.cpu().double() - wantW).abs().max()) <= 2e-5 * float(wantW.abs().max())
        assert float((db.cpu().double() - dout.double().sum(1)).abs().max()) <= 2e-5 * float(dout.double().sum(1).abs().max())
        dx = torch.empty(G, B, IN, device=dev)
        C.slv_heads_linear_bwd_x(ptr(dd), Wt.p, ptr(md) if use_mask else 0, msc, ptr(dx), G, B, IN, OUT, stream())
        wantx = torch.stack([(dout[gi].double() @ Ws[gi].double()) * (mask[gi].double() * msc if use_mask else 1.0) for gi in range(G)])
        assert float((dx.cpu().double() - wantx).abs().max()) <= 2e-5 * float(wantx.abs().max())


def test_gradients_against_the_fp64_oracle_on_a_well_conditioned_network():
    """A gradient check 10-1000x tighter than the noise-scaled ones above (those are held to the reference's fp32-vs-fp64
    deviation at ITS initialisation, ~1e-2 per tensor, and would not see a 1 % systematic error in one small tensor).  Here
    the residual blocks of both trunks start close to the identity (last BatchNorm gamma 0.1, the zero-init-residual recipe)
    and every BatchNorm sees >= 128 values per channel, which removes the BatchNorm amplification: every parameter gradient
    of the HIP step against the CPU oracle run in FLOAT64 -- per tensor the relative L2 error and the projection
    <g_hip, g_64> / |g_64|^2, plus fp64-accumulated directional derivatives sum_t <g_t - g64_t, d_t> along 8 random
    directions over ALL parameters (each tensor's direction scaled to its gradient's rms, so every tensor weighs the same).
    What remains irreducible in ANY fp32 implementation is the ReLU masks: a rounding flips the mask of the elements within
    1e-7 of zero, and the gradient's L2 error goes with the SQUARE ROOT of the flipped fraction -- the oracle's own fp32 run
    (torch CPU, measured in this test) sits at 3e-4 .. 1.6e-3 on the video trunk's tensors, 1.5e-4 in the projection and up
    to 7e-4 in the directional derivatives, but at ~1e-6 on the audio trunk and the heads and in the median.  So: tensors of a
    stage whose oracle-fp32 error is at rounding level are held to 2e-5 absolute; the others to 3 x the worst oracle-fp32
    error of their stage (<= 5e-3: a 1 % error in any one tensor fails); the median over all tensors to 1e-5."""
    hc, K, B = 2, 7, 8
    video = portable_fill_(torch.empty(B, 3, 8, 64, 64), 5)
    audio = portable_fill_(torch.empty(B, 1, 80, 64), 6)
    sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64))
    sel = torch.tensor([3, 17, 42, 63, 5, 9, 30, 51])

    def damp(m):
        with torch.no_grad():
            for li in range(1, 5):
                for blk in getattr(m.video_network.base, f"layer{li}"):
                    blk.conv2[1].weight.fill_(0.1)
                for blk in getattr(m.audio_network.base, f"layer{li}"):
                    blk.bn2.weight.fill_(0.1)

    grads = {}
    for kind in ("hip", "oracle64", "oracle32"):
        mod = model_ref if kind != "hip" else __import__("selavi_amd.model", fromlist=["x"])
        m = mod.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        portable_init_(m, seed=31)
        step_ref.set_dropout_p(m, 0.0)
        damp(m)
        if kind == "hip":
            from selavi_amd.utils import get_loss
            m = m.cuda().train()
            fv, fa = m(video.cuda(), audio.cuda())
            lab = sl.cuda()[sel.cuda(), :]
            loss = 0.5 * get_loss(fv, lab, hc) + 0.5 * get_loss(fa, lab, hc)
        else:
            dt = torch.float64 if kind == "oracle64" else torch.float32
            m = m.to(dt).train()
            fv, fa = m(video.to(dt), audio.to(dt))
            lab = sl[sel, :]
            loss = 0.5 * model_ref.get_loss(fv, lab, hc) + 0.5 * model_ref.get_loss(fa, lab, hc)
        loss.backward()
        grads[kind] = ({n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}, float(loss))
    g64, l64 = grads["oracle64"]
    assert set(g64) == set(grads["hip"][0])
    names = sorted(g64)
    rows = {}
    for kind in ("hip", "oracle32"):
        g, l = grads[kind]
        assert abs(l - l64) <= 1e-5 * abs(l64), (kind, l, l64)
        rel = np.array([float((g[n] - g64[n]).norm() / (g64[n].norm() + 1e-300)) for n in names])
        proj = np.array([float((g[n] * g64[n]).sum() / ((g64[n] * g64[n]).sum() + 1e-300)) for n in names])
        gen = torch.Generator().manual_seed(77)
        dd = []
        for _ in range(8):
            num = den = 0.0
            for n in names:
                d = torch.randn(g64[n].shape, generator=gen, dtype=torch.float64) / (g64[n].norm() / g64[n].numel() ** 0.5 + 1e-300)
                num += float(((g[n] - g64[n]) * d).sum())
                den += float(g64[n].numel())             # E <g, d>^2 = |g|^2 |d|^2 / n = n per tensor with this scaling
            dd.append(abs(num) / den ** 0.5)
        rows[kind] = (rel, proj, np.array(dd))
        print(f"{kind}: per-tensor rel L2 median {np.median(rel):.2e} worst {rel.max():.2e} ({names[int(rel.argmax())]}); "
              f"|proj - 1| worst {np.abs(proj - 1).max():.2e}; directional derivatives worst {max(dd):.2e}")
    rel, proj, dd = rows["hip"]
    rel32, proj32, dd32 = rows["oracle32"]

    def stage(n):
        p = n.split(".")
        return ".".join(p[:3]) if p[0].endswith("_network") else p[0][:5]      # video layerN / stem, audio layerN / conv1, heads
    worst32, wproj32 = {}, {}
    for n, r, pj in zip(names, rel32, proj32):
        worst32[stage(n)] = max(worst32.get(stage(n), 0.0), r)
        wproj32[stage(n)] = max(wproj32.get(stage(n), 0.0), abs(pj - 1))
    bad = [(n, r, max(2e-5, 3 * worst32[stage(n)])) for n, r in zip(names, rel) if r > max(2e-5, 3 * worst32[stage(n)])]
    assert not bad, bad[:8]
    badp = [(n, pj) for n, pj in zip(names, proj) if abs(pj - 1) > max(2e-5, 3 * wproj32[stage(n)])]
    assert not badp, badp[:8]
    assert np.median(rel) <= 1e-5 and np.median(np.abs(proj - 1)) <= 1e-6, (np.median(rel), np.median(np.abs(proj - 1)))
    assert dd.max() <= max(1e-4, 3 * dd32.max()), (dd, dd32)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_batched_weight_images_leave_the_step_bit_identical(monkeypatch, precision):
    """ops.WeightImages: from the second pass on every split-operand weight image of a trunk is made by ONE launch at the head
    of the trunk (slv_conv_w_transform_jobs) instead of one launch per conv layer -- same kernels' arithmetic, same images:
    four training steps and an eval pass with the batching on and off end in bit-identical weights, buffers and features."""
    from selavi_amd import ops, optim, train

    def run(flag):
        monkeypatch.setattr(ops, "BATCH_W_IMAGES", flag)
        m = _build(2, 12, True).train()
        m.set_precision(precision)
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
        audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
        sl = torch.from_numpy((np.arange(64 * 2).reshape(64, 2) * 7919 % 12).astype(np.int64)).cuda()
        sel = torch.tensor([3, 17, 42, 63]).cuda()
        losses = [float(train.train_step(m, opt, video, audio, sl, sel, 2)) for _ in range(4)]
        m.eval()
        m.return_features = True
        with torch.no_grad():
            f1 = m(video, audio)
            f2 = m(video, audio)            # (the second eval pass runs on the batched images)
        assert torch.equal(f1[0], f2[0]) and torch.equal(f1[1], f2[1])
        w = m.video_network.base.__dict__.get("_wimg_cur")
        return losses, {k: v.clone() for k, v in m.state_dict().items()}, f2, w
    l1, s1, f1, w1 = run(True)
    l0, s0, f0, w0 = run(False)
    assert w0 is None and w1 is not None and w1.ready and w1.njobs >= 20, (w1 and w1.njobs)
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    assert torch.equal(f0[0], f1[0]) and torch.equal(f0[1], f1[1])
