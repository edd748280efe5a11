"""Diagnostic (GPU): per-parameter gradient error of the HIP engine against the committed fp64 golden
gradients of the executed reference (tests/golden/grads_hc1_k28.npz), next to the reference's own
fp32-vs-fp64 deviation e_cpu.  Needs no oracle run.  Usage: python tests/diag/grad_vs_golden.py [top]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle.model_ref import portable_fill_, portable_init_
from oracle import step_ref
from selavi_amd import model as smodel
from selavi_amd.utils import get_loss

top = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gd, "model_hc1_k28_mlp1.npz"))
gg = np.load(os.path.join(gd, "grads_hc1_k28.npz"))
hc, K = int(g["hc"]), int(g["K"])
B, T, S = int(g["B"]), int(g["T"]), int(g["S"])
m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
portable_init_(m, seed=31)
step_ref.set_dropout_p(m, 0.0)
m = m.cuda().train()
video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
audio = portable_fill_(torch.empty(B, 1, 40, 36), 6).cuda()
sl = torch.from_numpy(g["selflabels"]).cuda()
sel = torch.from_numpy(g["selected"]).cuda()
fv, fa = m(video, audio)
labels = sl[sel, 0]
loss = 0.5 * get_loss(fv, labels, hc) + 0.5 * get_loss(fa, labels, hc)
loss.backward()
names = [str(n) for n in gg["names"]]
params = dict(m.named_parameters())
rows = []
for i, name in enumerate(names):
    gr = params[name].grad.detach().double().cpu()
    n = min(256, gr.numel())
    ref_norm = float(gg["norms"][i])
    err = (gr.flatten()[:n] - torch.from_numpy(gg["heads"][i, :n])).norm().item()
    scale = ref_norm * (n / gr.numel()) ** 0.5 + 1e-30
    rows.append((err / scale, float(gg["e_cpu"][i]), abs(gr.norm().item() - ref_norm) / ref_norm, name))
rows.sort(reverse=True)
print("loss %.8f   (err/rms, reference fp32 noise e_cpu, norm rel err, tensor)" % loss.item())
for r in rows[:top]:
    print("%.2e  %.2e  %.2e  %s" % r)
print("median err %.2e   median e_cpu %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
