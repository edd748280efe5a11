"""Diagnostic: per-parameter gradient error of the HIP engine vs an fp64 CPU oracle, next to the
fp32 CPU oracle's own error (noise floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import model_ref, step_ref
from oracle.model_ref import portable_fill_, portable_init_
from selavi_amd import model as smodel
from selavi_amd.utils import get_loss

hc, K, use_mlp = int(sys.argv[1]), int(sys.argv[2]), True
B, T, S = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.set_num_threads(os.cpu_count())
video = portable_fill_(torch.empty(B, 3, T, S, S), 5)
audio = portable_fill_(torch.empty(B, 1, 40, 50), 6)
N = 64
sl = torch.from_numpy((np.arange(N * hc).reshape(N, hc) * 7919 % K).astype(np.int64))
sel = torch.arange(B) * 3

def oracle(dtype):
    o = model_ref.load_model(use_mlp=use_mlp, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(o, seed=31); step_ref.set_dropout_p(o, 0.0)
    o = o.to(dtype).train()
    fv, fa = o(video.to(dtype), audio.to(dtype))
    labels = sl[sel, 0] if hc == 1 else sl[sel, :]
    loss = 0.5 * model_ref.get_loss(fv, labels, hc) + 0.5 * model_ref.get_loss(fa, labels, hc)
    loss.backward()
    return loss.item(), {k: p.grad.detach().double() for k, p in o.named_parameters()}

l64, g64 = oracle(torch.float64)
l32, g32 = oracle(torch.float32)
m = smodel.load_model(use_mlp=use_mlp, num_classes=K, norm_feat=False, headcount=hc)
portable_init_(m, seed=31); step_ref.set_dropout_p(m, 0.0)
m = m.cuda().train()
fv, fa = m(video.cuda(), audio.cuda())
labels = (sl[sel, 0] if hc == 1 else sl[sel, :]).cuda()
loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
loss.backward()
print("loss fp64 %.8f fp32cpu %.8f hip %.8f" % (l64, l32, loss.item()))
rows = []
for name, p in m.named_parameters():
    ref = g64[name]; nrm = ref.norm().item() + 1e-30
    rows.append(((p.grad.cpu().double() - ref).norm().item() / nrm, (g32[name] - ref).norm().item() / nrm, name))
rows.sort(reverse=True)
for e_hip, e_cpu, name in rows[:15]:
    print("%.2e  %.2e  %s" % (e_hip, e_cpu, name))
print("median hip %.2e cpu %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
