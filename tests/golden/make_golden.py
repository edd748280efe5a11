#!/usr/bin/env python
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE in the build container.

Runs only where ``/root/reference`` exists (never on the GPU box).  It imports the reference's
own ``model.py``, ``utils.py`` and ``src/sk_utils.py`` unmodified, behind two container-only shims
(SURVEY.md section 8c):
  1. a ``sys.modules['torchvision']`` stand-in whose networks are ``oracle.model_ref``'s restatement
     of torchvision 0.4.2 (torchvision is not vendored in the reference nor installed here);
  2. a cuda->cpu remap for ``src/sk_utils.py``'s hard-coded ``device='cuda'`` / ``.cuda()``.
Only input/output *data* is written to the fixtures; no reference source is copied.

    python tests/golden/make_golden.py
"""
import hashlib
import logging
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from oracle import model_ref  # noqa: E402
from oracle.model_ref import portable_fill_, portable_init_  # noqa: E402


# ------------------------------------------------------------------ shims (container only)
def install_torchvision_standin():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    video = types.ModuleType("torchvision.models.video")
    resnet = types.ModuleType("torchvision.models.resnet")
    video.r2plus1d_18 = model_ref.r2plus1d_18
    resnet._resnet = model_ref._resnet
    resnet.BasicBlock = model_ref.BasicBlock
    resnet.resnet18 = model_ref.resnet18
    models.video, models.resnet, models.resnet18 = video, resnet, model_ref.resnet18
    models.resnet34, models.resnet50 = model_ref.resnet34, model_ref.resnet50      # model.py:106 torchvision.models.__dict__[arch]
    resnet.resnet34, resnet.resnet50, resnet.Bottleneck = model_ref.resnet34, model_ref.resnet50, model_ref.Bottleneck
    tv.models = models
    for name, m in [("torchvision", tv), ("torchvision.models", models),
                    ("torchvision.models.video", video), ("torchvision.models.resnet", resnet)]:
        sys.modules[name] = m


def install_cuda_to_cpu():
    def wrap(fn):
        def inner(*a, **k):
            if 'device' in k and str(k['device']).startswith('cuda'):
                k['device'] = 'cpu'
            return fn(*a, **k)
        return inner
    for name in ("ones", "zeros", "randn", "arange", "empty"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None


def import_reference():
    install_torchvision_standin()
    install_cuda_to_cpu()
    sys.path.insert(0, REF)
    import model as ref_model          # /root/reference/model.py
    import utils as ref_utils          # /root/reference/utils.py
    from src import sk_utils as ref_sk  # /root/reference/src/sk_utils.py
    return ref_model, ref_utils, ref_sk


class Args:
    """The SK fields of opt.py the solver reads (sk_utils.py:367-388)."""
    def __init__(self, **kw):
        self.distribution, self.dist, self.diff_dist_every = 'default', None, False
        self.diff_dist_per_head, self.gauss_sd, self.headcount = True, 0.1, 1
        self.lamb, self.rank = 20, 0
        self.__dict__.update(kw)


def synth_PS(N, K, scale, seed):
    """PS = softmax64(s*G1) * softmax64(s*G2), G ~ portable N(0,1)   (SURVEY 8d)."""
    g1 = portable_fill_(torch.empty(N, K, dtype=torch.float64), seed, kind="normal").numpy()
    g2 = portable_fill_(torch.empty(N, K, dtype=torch.float64), seed + 1, kind="normal").numpy()
    from oracle.sk_ref import softmax64
    return softmax64(scale * g1) * softmax64(scale * g2)


def label_digest(L):
    return hashlib.sha256(np.ascontiguousarray(L.astype(np.int32)).tobytes()).hexdigest()


def grad_fixture(ref_model, ref_utils, fname="grads_hc1_k28.npz", B=4, T=4, S=32, FA=40, TA=36):
    """All-parameter gradient fixture: fp64 run of the reference model (truth), fp32 run (the
    reference's own arithmetic noise), first 256 elements + norm of every parameter gradient.
    The default (tiny) shapes leave 16 elements per channel in layer4's BatchNorms: gradients there are
    ill-conditioned (1e-7 perturbations show up as 1e-2).  The "wide" variant (B=6, T=8, S=64, 96x96
    audio) keeps >= 54 elements per channel and is the tight parity fixture."""
    hc, K = 1, 28
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5, kind="normal")
    audio = portable_fill_(torch.empty(B, 1, FA, TA), 6, kind="normal")
    N = 64
    selflabels = torch.from_numpy((np.arange(N * hc).reshape(N, hc) * 7919 % K).astype(np.int64))
    selected = torch.tensor([3, 17, 42, 63, 8, 29][:B]) if B <= 6 else (torch.arange(B) * 4 + 1)
    res = {}
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        m = ref_model.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True,
                                 num_classes=K, pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc)
        portable_init_(m, seed=31)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m = m.to(dtype).train()
        fv, fa = m(video.to(dtype), audio.to(dtype))
        labels = selflabels[selected, 0]
        loss = 0.5 * ref_utils.get_loss(fv, labels, headcount=hc) + 0.5 * ref_utils.get_loss(fa, labels, headcount=hc)
        loss.backward()
        res[tag] = {k: p.grad.detach().double() for k, p in m.named_parameters()}
        res[tag + "_loss"] = loss.item()
    names = list(res["f64"].keys())
    heads = np.zeros((len(names), 256))
    norms = np.zeros(len(names))
    e_cpu = np.zeros(len(names))
    for i, k in enumerate(names):
        g64, g32 = res["f64"][k].flatten(), res["f32"][k].flatten()
        n = min(256, g64.numel())
        heads[i, :n] = g64[:n].numpy()
        norms[i] = g64.norm().item()
        e_cpu[i] = (g32 - g64).norm().item() / (norms[i] + 1e-300)
    np.savez_compressed(os.path.join(OUT, fname), names=np.array(names), heads=heads, norms=norms,
                        e_cpu=e_cpu, loss64=res["f64_loss"], loss32=res["f32_loss"], hc=hc, K=K, B=B, T=T, S=S,
                        FA=FA, TA=TA,
                        selflabels=selflabels.numpy(), selected=selected.numpy())
    print("grad fixture: median fp32 noise %.2e max %.2e" % (np.median(e_cpu), e_cpu.max()))


def cfg1_fixture(ref_model, ref_utils):
    """BASELINE configs[0] at FULL size (the reference's own CPU-runnable case): bs=4, 8x112x112 clips,
    1x40x100 log-mel, K=28, headcount=1.  Eval and train-mode logits, trunk features and the loss of
    main.py:284-293, produced by the executed reference."""
    hc, K, B = 1, 28, 4
    m = ref_model.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=K,
                             pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc)
    portable_init_(m, seed=31)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    video = portable_fill_(torch.empty(B, 3, 8, 112, 112), 45, kind="normal")
    audio = portable_fill_(torch.empty(B, 1, 40, 100), 46, kind="normal")
    selflabels = torch.from_numpy((np.arange(3328) * 7919 % K).astype(np.int64)).view(-1, 1)
    selected = torch.tensor([3, 1700, 42, 3327])
    out = {}
    m.eval()
    with torch.no_grad():
        fv, fa = m(video, audio)
        m.return_features = True
        gv, ga = m(video, audio)
        m.return_features = False
    out["eval_v"], out["eval_a"], out["feat_v"], out["feat_a"] = fv.numpy(), fa.numpy(), gv.numpy(), ga.numpy()
    m.train()
    fv, fa = m(video, audio)
    labels = selflabels[selected, 0]
    loss = 0.5 * ref_utils.get_loss(fv, labels, headcount=hc) + 0.5 * ref_utils.get_loss(fa, labels, headcount=hc)
    out["train_v"], out["train_a"], out["loss"] = fv.detach().numpy(), fa.detach().numpy(), np.float64(loss.item())
    sd = m.state_dict()
    for key in ("video_network.base.layer4.1.conv2.1.running_var", "audio_network.base.bn1.running_mean"):
        out["post/" + key] = sd[key].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "cfg1_full.npz"), hc=hc, K=K, B=B, selflabels=selflabels.numpy(),
                        selected=selected.numpy(), **out)
    print("cfg1 full-size fixture: loss", loss.item())


def cfg2_fixture(ref_model, ref_utils, hc=10, K=309, B=16, T=16, fname="cfg2_full.npz"):
    """BASELINE configs[1] at FULL size (the configuration the metric is quoted on): bs=16, 16x112x112
    clips, 1x129x100 log-mel, K=309, headcount=10.  Train-mode (batch statistics) trunk features, the
    logits of heads 0 and 9 of both modalities and the loss of main.py:284-293 from the executed reference.
    Also used for configs[3] (Kinetics-400 shape: 30 frames -> odd temporal sizes 30/15/8/4, K=400)."""
    m = ref_model.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=K,
                             pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc)
    portable_init_(m, seed=31)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    video = portable_fill_(torch.empty(B, 3, T, 112, 112), 55, kind="normal")
    audio = portable_fill_(torch.empty(B, 1, 129, 100), 56, kind="normal")
    selflabels = torch.from_numpy((np.arange(1024 * hc).reshape(1024, hc) * 7919 % K).astype(np.int64))
    selected = torch.arange(B) * 61
    m.train()
    with torch.no_grad():
        fv, fa = m(video, audio)
        labels = selflabels[selected, :]
        loss = 0.5 * ref_utils.get_loss(fv, labels, headcount=hc) + 0.5 * ref_utils.get_loss(fa, labels, headcount=hc)
    out = dict(train_v0=fv[0].numpy(), train_v9=fv[9].numpy(), train_a0=fa[0].numpy(), train_a9=fa[9].numpy(),
               loss=np.float64(loss.item()))
    m.eval()
    m.return_features = True
    with torch.no_grad():
        gv, ga = m(video, audio)       # eval-mode features with the running statistics of ONE train-mode forward
    out["feat_v"], out["feat_a"] = gv.numpy(), ga.numpy()
    np.savez_compressed(os.path.join(OUT, fname), hc=hc, K=K, B=B, T=T, selflabels=selflabels.numpy(),
                        selected=selected.numpy(), **out)
    print(fname, "full-size fixture: loss", loss.item())


def audio_archs_fixture(ref_model, fname="audio_archs.npz"):
    """The audio trunks model.py:103-110 accepts beside resnet9 -- resnet18 / resnet34 / resnet50 -- through the EXECUTED
    reference's own load_model (get_audio_feature_extractor: torchvision.models.__dict__[arch], conv1 replaced by a
    1-channel 7x7 conv AFTER the init, fc -> Identity), tiny shapes: eval- and train-mode logits of a 2-head model for the
    512-d trunks, eval- and train-mode trunk features for all three (resnet50's 2048 features do not fit the 512-d heads the
    reference builds: return_features only), and the running statistics one train-mode forward leaves behind."""
    hc, K, B = 2, 12, 4
    video = portable_fill_(torch.empty(B, 3, 4, 32, 32), 5, kind="normal")
    audio = portable_fill_(torch.empty(B, 1, 80, 64), 6, kind="normal")
    out = {}
    for arch in ("resnet18", "resnet34", "resnet50"):
        m = ref_model.load_model(vid_base_arch='r2plus1d_18', aud_base_arch=arch, use_mlp=True, num_classes=K,
                                 pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc)
        portable_init_(m, seed=31)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        heads = arch != "resnet50"
        m.eval()
        with torch.no_grad():
            m.return_features = True
            _, ga = m(video, audio)
            m.return_features = False
            out[arch + "/eval_feat_a"] = ga.numpy()
            if heads:
                _, fa = m(video, audio)
                out[arch + "/eval_a"] = np.stack([t.numpy() for t in fa])
        m.train()
        with torch.no_grad():
            m.return_features = True
            _, ga = m(video, audio)
            m.return_features = False
            out[arch + "/train_feat_a"] = ga.numpy()
        sd = m.state_dict()
        last = "audio_network.base.layer4.%d." % (len(m.audio_network.base.layer4) - 1)
        key = last + ("bn3" if arch == "resnet50" else "bn2") + ".running_var"
        out[arch + "/post_running_var"] = sd[key].numpy().copy()
        out[arch + "/n_keys_audio"] = len([k for k in sd if k.startswith("audio_network.")])
        out[arch + "/n_params_audio"] = sum(p.numel() for p in m.audio_network.parameters())
        print("audio arch fixture", arch, "features", ga.shape, "params", out[arch + "/n_params_audio"])
    np.savez_compressed(os.path.join(OUT, fname), hc=hc, K=K, B=B, **out)


def sk_kinetics_fixture(ref_sk, name="sk_kinetics_full", N=230976, K=400, scale=1.0, seed=41, hc=2, head=1):
    """BASELINE configs[3]'s Sinkhorn-Knopp problem solved by the reference itself (sk_utils.py:359-422 with the gauss
    branch :368-388, cluster sizes per head GIVEN -- torch's randn stream does not travel): digest / histogram / head and
    tail of the labels, cost; iteration count and alpha from the oracle after it reproduced the reference's labels."""
    from oracle import sk_ref
    PS = synth_PS(N, K, scale, seed)
    dists = [(portable_fill_(torch.empty(K, 1, dtype=torch.float64), 200 + h, kind="normal") * 0.1 + 1.0) * N / K
             for h in range(hc)]
    dist_in = np.stack([d.numpy().ravel() for d in dists])
    args = Args(distribution='gauss', dist=[d.clone() for d in dists], headcount=hc)
    cost, newL = ref_sk.optimize_L_sk_gpu(args, torch.from_numpy(PS.copy()), head, logging.getLogger("golden"))
    newL = newL.numpy()
    kd = sk_ref.marginals(K, N, PS, 'gauss', dist_in[head])
    cost_o, L_o, info = sk_ref.optimize_L_sk(PS, lamb=20, K_dist=kd)
    assert (L_o == newL).all()
    assert abs(cost_o - cost) <= 1e-12 * abs(cost), (cost_o, cost)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), N=N, K=K, scale=scale, seed=seed, lamb=20, cost=cost, iters=info["iters"],
        alpha=info["alpha"], labels=np.zeros(0, np.int16), labels_head=newL[:4096].astype(np.int16),
        labels_tail=newL[-4096:].astype(np.int16), hist=np.bincount(newL, minlength=K).astype(np.int32),
        digest=np.frombuffer(label_digest(newL).encode(), dtype=np.uint8), dist_in=dist_in,
        dist_after=np.stack([d.numpy().ravel() for d in args.dist]), head=head)
    print(name, "iters", info["iters"], "cost", cost)


def main():
    if "--only-cfg1" in sys.argv or "--only-cfg2" in sys.argv:
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        (cfg1_fixture if "--only-cfg1" in sys.argv else cfg2_fixture)(ref_model, ref_utils)
        return
    if "--only-audio-archs" in sys.argv:      # model.py:103-110: resnet18 / resnet34 / resnet50 audio trunks (seconds of CPU)
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        audio_archs_fixture(ref_model)
        return
    if "--only-sk-kinetics" in sys.argv:      # configs[3]: N = 230 976, K = 400, gauss marginals per head, the reference on CPU (~minutes)
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        sk_kinetics_fixture(ref_sk)
        return
    if "--only-cfg4" in sys.argv:             # configs[3]: 30-frame clips, K=400, hc=10 (per-GPU bs 16 as in scripts/master.sh:82; ~2 min of CPU)
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        cfg2_fixture(ref_model, ref_utils, hc=10, K=400, B=16, T=30, fname="cfg4_full.npz")
        return
    if "--only-cfg2-grads" in sys.argv:       # full-size (bs 16, 16x112x112) fp64 + fp32 gradients: ~10 min of CPU
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        grad_fixture(ref_model, ref_utils, "grads_cfg2_full.npz", B=16, T=16, S=112, FA=129, TA=100)
        return
    if "--only-grads" in sys.argv:
        ref_model, ref_utils, ref_sk = import_reference()
        torch.set_num_threads(os.cpu_count())
        grad_fixture(ref_model, ref_utils)
        grad_fixture(ref_model, ref_utils, "grads_hc1_k28_wide.npz", B=6, T=8, S=64, FA=96, TA=96)
        return
    torch.manual_seed(0)
    np.random.seed(0)
    ref_model, ref_utils, ref_sk = import_reference()
    logger = logging.getLogger("golden")
    torch.set_num_threads(os.cpu_count())

    # ---------------------------------------------------------------- SK fixtures
    sk_cases = [
        dict(name="sk_ave_uniform", N=3328, K=28, scale=1.0, seed=11, distribution='default'),
        dict(name="sk_ave_peaked", N=3328, K=28, scale=4.0, seed=13, distribution='default'),
        dict(name="sk_k309_small", N=4096, K=309, scale=4.0, seed=17, distribution='default'),
        dict(name="sk_k400_ragged", N=1037, K=400, scale=2.0, seed=19, distribution='default'),
        dict(name="sk_gauss_per_head", N=3328, K=28, scale=2.0, seed=23, distribution='gauss'),
        dict(name="sk_vggsound_full", N=170752, K=309, scale=1.0, seed=29, distribution='default'),
    ]
    for cs in sk_cases:
        N, K = cs["N"], cs["K"]
        PS = synth_PS(N, K, cs["scale"], cs["seed"])
        extra = {}
        if cs["distribution"] == 'gauss':
            hc = 3
            dists = [portable_fill_(torch.empty(K, 1, dtype=torch.float64), 100 + h, kind="normal")
                     * 0.1 + 1.0 for h in range(hc)]
            dists = [d * N / K for d in dists]                        # sk_utils.py:372 form, given
            extra["dist_in"] = np.stack([d.numpy().ravel() for d in dists])
            args = Args(distribution='gauss', dist=[d.clone() for d in dists], headcount=hc)
            head = 1
        else:
            args, head = Args(), 0
        cost, newL = ref_sk.optimize_L_sk_gpu(args, torch.from_numpy(PS.copy()), head, logger)
        # iteration count / alpha are not returned by the reference; recover them with the oracle
        from oracle import sk_ref
        kd = None
        if cs["distribution"] == 'gauss':
            kd = sk_ref.marginals(K, N, PS, 'gauss', extra["dist_in"][head])
            extra["dist_after"] = np.stack([d.numpy().ravel() for d in args.dist])  # mutated in place
            extra["head"] = head
        cost_o, L_o, info = sk_ref.optimize_L_sk(PS, lamb=20, K_dist=kd)
        newL = newL.numpy()
        assert (L_o == newL).all(), cs["name"]
        assert abs(cost_o - cost) <= 1e-12 * abs(cost), (cost_o, cost)
        full = N * K <= 4096 * 400
        np.savez_compressed(
            os.path.join(OUT, cs["name"] + ".npz"),
            N=N, K=K, scale=cs["scale"], seed=cs["seed"], lamb=20, cost=cost, iters=info["iters"],
            alpha=info["alpha"], labels=newL.astype(np.int16) if full else np.zeros(0, np.int16),
            labels_head=newL[:4096].astype(np.int16), labels_tail=newL[-4096:].astype(np.int16),
            hist=np.bincount(newL, minlength=K).astype(np.int32),
            digest=np.frombuffer(label_digest(newL).encode(), dtype=np.uint8), **extra)
        print(cs["name"], "iters", info["iters"], "cost", cost)

    # ---------------------------------------------------------------- model / loss / step fixtures
    ref_model_mod = ref_model
    for hc, K, use_mlp in [(1, 28, True), (3, 12, True), (2, 7, False)]:
        torch.manual_seed(1)
        m = ref_model_mod.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=use_mlp,
                                     num_classes=K, pretrained=False, norm_feat=False, use_max_pool=False,
                                     headcount=hc)
        portable_init_(m, seed=31)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0                       # RNG streams are not portable
        B, T, S = 4, 4, 32
        video = portable_fill_(torch.empty(B, 3, T, S, S), 5, kind="normal")
        audio = portable_fill_(torch.empty(B, 1, 40, 36), 6, kind="normal")
        N = 64
        selflabels = torch.from_numpy((np.arange(N * hc).reshape(N, hc) * 7919 % K).astype(np.int64))
        selected = torch.tensor([3, 17, 42, 63])
        out = {}
        # eval-mode forward with fresh running stats
        m.eval()
        with torch.no_grad():
            fv, fa = m(video, audio)
            m.return_features = True
            gv, ga = m(video, audio)
            m.return_features = False
        out["eval_v"] = np.stack([t.numpy() for t in (fv if hc > 1 else [fv])])
        out["eval_a"] = np.stack([t.numpy() for t in (fa if hc > 1 else [fa])])
        out["feat_v"], out["feat_a"] = gv.numpy(), ga.numpy()
        # two training steps with the reference's loss + torch SGD (main.py:132-137,284-302)
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        losses = []
        for step in range(2):
            fv, fa = m(video, audio)
            labels = selflabels[selected, 0] if hc == 1 else selflabels[selected, :]
            loss = 0.5 * ref_utils.get_loss(fv, labels, headcount=hc) + \
                0.5 * ref_utils.get_loss(fa, labels, headcount=hc)
            if step == 0:
                out["train_v"] = np.stack([t.detach().numpy() for t in (fv if hc > 1 else [fv])])
                out["train_a"] = np.stack([t.detach().numpy() for t in (fa if hc > 1 else [fa])])
            opt.zero_grad()
            loss.backward()
            if step == 0:
                sd = dict(m.named_parameters())
                for key in ["video_network.base.stem.0.weight", "video_network.base.layer4.1.conv2.0.3.weight",
                            "video_network.base.layer2.0.downsample.0.weight",
                            "video_network.base.layer1.0.conv1.0.1.weight",
                            "audio_network.base.conv1.weight", "audio_network.base.layer3.0.bn2.bias"]:
                    g = sd[key].grad
                    out["gradnorm/" + key] = np.float64(g.double().norm().item())
                    out["gradhead/" + key] = g.flatten()[:64].numpy().copy()
            opt.step()
            losses.append(loss.item())
        out["losses"] = np.array(losses)
        sdict = m.state_dict()
        for key in ["video_network.base.stem.0.weight", "video_network.base.layer3.1.conv1.0.0.weight",
                    "video_network.base.stem.1.running_mean", "video_network.base.stem.1.running_var",
                    "video_network.base.layer4.1.conv2.1.running_var",
                    "audio_network.base.bn1.running_mean", "audio_network.base.layer4.0.conv2.weight"]:
            out["post/" + key] = sdict[key].flatten()[:64].numpy().copy()
        out["n_keys"] = len(sdict)
        out["n_params"] = sum(p.numel() for p in m.parameters())
        np.savez_compressed(os.path.join(OUT, f"model_hc{hc}_k{K}_mlp{int(use_mlp)}.npz"),
                            hc=hc, K=K, use_mlp=use_mlp, B=B, T=T, S=S, selflabels=selflabels.numpy(),
                            selected=selected.numpy(), **out)
        print("model fixture", hc, K, use_mlp, "losses", losses, "keys", len(sdict))

    cfg1_fixture(ref_model_mod, ref_utils)
    cfg2_fixture(ref_model_mod, ref_utils)
    audio_archs_fixture(ref_model_mod)

    # state-dict key lists of the full-size configs (cfg1 hc=1, cfg2 hc=10): names only
    for hc, K in [(1, 28), (10, 309)]:
        m = ref_model_mod.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        keys = list(m.state_dict().keys())
        shapes = [tuple(v.shape) for v in m.state_dict().values()]
        with open(os.path.join(OUT, f"state_dict_keys_hc{hc}.txt"), "w") as f:
            for k, s in zip(keys, shapes):
                f.write(f"{k} {list(s)}\n")
        print("keys", hc, len(keys), sum(p.numel() for p in m.parameters()))

    grad_fixture(ref_model_mod, ref_utils)

    # get_loss fixture (utils.py:377-387)
    acts = [portable_fill_(torch.empty(5, 9), 40 + h, kind="normal") for h in range(3)]
    tg = torch.from_numpy((np.arange(15).reshape(5, 3) * 5 % 9).astype(np.int64))
    np.savez_compressed(os.path.join(OUT, "get_loss.npz"), acts=np.stack([a.numpy() for a in acts]),
                        targets=tg.numpy(),
                        loss_hc3=ref_utils.get_loss(acts, tg, headcount=3).item(),
                        loss_hc1=ref_utils.get_loss(acts[0], tg[:, 0], headcount=1).item())

    # match_order fixture (sk_utils.py:424-467): record np.random.choice's pair sequence
    K, N = 12, 257
    e1 = synth_PS(N, K, 2.0, 51)
    perm_true = np.random.RandomState(3).permutation(K)
    e2 = e1[:, perm_true] + 1e-3 * synth_PS(N, K, 1.0, 53)
    W2 = torch.nn.Linear(16, K)
    portable_fill_(W2.weight.data, 61)
    portable_fill_(W2.bias.data, 62)
    w_before, b_before = W2.weight.data.numpy().copy(), W2.bias.data.numpy().copy()
    pairs = []
    orig_choice = np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        pairs.append(np.array(r))
        return r
    np.random.choice = rec_choice
    import torch.distributed as dist
    dist.broadcast = lambda t, src, *a, **k: None
    np.random.seed(5)
    ref_sk.match_order(Args(), torch.from_numpy(e1), torch.from_numpy(e2), W2, steps=3000, restarts=2,
                       logger=logger)
    np.random.choice = orig_choice
    np.savez_compressed(os.path.join(OUT, "match_order.npz"), emb1=e1, emb2=e2, pairs=np.stack(pairs),
                        w_before=w_before, b_before=b_before, w_after=W2.weight.data.numpy(),
                        b_after=W2.bias.data.numpy())
    print("match_order pairs", len(pairs))


if __name__ == "__main__":
    main()
