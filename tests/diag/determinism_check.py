"""Run-to-run determinism of the training step (GPU): N repetitions of {2 steps from the same state}
must give bit-identical parameters and losses.  Usage: python tests/diag/determinism_check.py [reps] [B T S]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle.model_ref import portable_fill_, portable_init_
from selavi_amd import model as smodel, optim, train

if os.environ.get("DET_POISON"):      # every fresh float allocation starts as NaN (or a given value): finds reads of unwritten memory
    _pv = float(os.environ["DET_POISON"])
    _e, _el = torch.empty, torch.empty_like
    def _pe(*a, **k):
        t = _e(*a, **k)
        return t.fill_(_pv) if t.is_floating_point() and t.is_cuda else t
    def _pel(*a, **k):
        t = _el(*a, **k)
        return t.fill_(_pv) if t.is_floating_point() and t.is_cuda else t
    torch.empty, torch.empty_like = _pe, _pel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, T, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 8, 64)
hc, K = 2, 31
m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
portable_init_(m, seed=31)
m = m.cuda().train()
state0 = {k: v.clone() for k, v in m.state_dict().items()}
video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
audio = portable_fill_(torch.empty(B, 1, 65, 50), 6).cuda()
selflabels = (torch.arange(64 * hc).view(64, hc) * 7 % K).cuda()
selected = torch.arange(B).cuda() * 3
sigs = []
for r in range(reps):
    m.load_state_dict(state0)
    torch.manual_seed(0)                       # dropout masks
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    SYNC = os.environ.get("DET_SYNC", "")       # debugging: device-wide syncs at chosen points (A,B,C)
    from selavi_amd.utils import get_loss
    losses = []
    if os.environ.get("DET_FWD"):               # forward only (train mode: batch statistics + running stats)
        with torch.no_grad():
            for _ in range(3):
                fv, fa = m(video, audio)
        losses = [float(sum(t.double().sum() for t in list(fv) + list(fa)))]
    for _ in range(0 if os.environ.get("DET_FWD") else 2):
        fv, fa = m(video, audio)
        labels = selflabels[selected, :]
        loss = 0.5 * get_loss(fv, labels, headcount=hc) + 0.5 * get_loss(fa, labels, headcount=hc)
        if "A" in SYNC: torch.cuda.synchronize()
        opt.zero_grad()
        loss.backward()
        if "B" in SYNC: torch.cuda.synchronize()
        opt.step()
        if "C" in SYNC: torch.cuda.synchronize()
        losses.append(float(loss))
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, v in m.state_dict().items():
        h.update(v.detach().cpu().numpy().tobytes())
    sigs.append((h.hexdigest()[:16], losses))
    print(r, sigs[-1])
    cur = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if r == 0:
        ref = cur
    else:
        bad = [(k, float((cur[k].double() - ref[k].double()).abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
        if bad:
            print("   differing tensors vs rep 0:", len(bad), bad[:6])
print("DETERMINISTIC" if len({s[0] for s in sigs}) == 1 else "NON-DETERMINISTIC")
