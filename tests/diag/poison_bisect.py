"""Which op's result depends on the contents of fresh allocations?  Stage-by-stage hashes of a train-mode and an
eval-mode forward (fp32 path) under DET_POISON (unset / 1e30 / nan).  Usage: DET_POISON=1e30 python tests/diag/poison_bisect.py"""
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
from oracle.model_ref import portable_fill_, portable_init_
from selavi_amd import engine, model as smodel, ops


def h(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().float().cpu().numpy()).tobytes()).hexdigest()[:8]


B, T, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 4, 32)
m = smodel.load_model(use_mlp=True, num_classes=8, norm_feat=False, headcount=1)
portable_init_(m, seed=31)
m = m.cuda()
video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
audio = portable_fill_(torch.empty(B, 1, 40, 36), 6).cuda()
base = m.video_network.base
for training in (True, False):
    ctx = engine.Ctx(training)
    x = video
    out = []
    with torch.no_grad():
        for st in engine.VIDEO_STAGES:
            x, saved = engine.video_stage_forward(ctx, base, st, x)
            out.append(f"{st}:{h(x)}")
        fa, _ = engine.audio_forward(engine.Ctx(training), m.audio_network.base, audio)
        out.append(f"audio:{h(fa)}")
        m.train(training)
        lv, la = m(video, audio)
        out.append(f"logits:{h(lv)},{h(la)}")
    rs = torch.cat([b.running_mean for b in m.modules() if hasattr(b, "running_mean")])
    print(f"poison {POISON} {'train' if training else 'eval '} " + " ".join(out) + f" running:{h(rs)}", flush=True)
