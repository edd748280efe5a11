"""Per-parameter agreement of the bf16-path gradients with the fp32 HIP step (diagnostic, not a test)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_train16_gpu import _step_setup
from selavi_amd.utils import get_loss

B, T, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res = {}
for prec in ("fp32", "bf16"):
    m, opt, video, audio, sl, sel, hc = _step_setup(prec, B=B, T=T, S=S)
    fv, fa = m(video, audio)
    labels = sl[sel, :]
    loss = 0.5 * get_loss(fv, labels, headcount=hc) + 0.5 * get_loss(fa, labels, headcount=hc)
    opt.zero_grad()
    loss.backward()
    res[prec] = (float(loss), {n: p.grad.detach().double().flatten() for n, p in m.named_parameters() if n.startswith("video")})
print("loss", res["fp32"][0], res["bf16"][0])
for n, a in res["fp32"][1].items():
    b = res["bf16"][1][n]
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    print(f"{cos:8.4f} rel {float((a-b).norm()/(a.norm()+1e-30)):8.4f} |g| {float(a.norm()):9.3e} {a.numel():8d} {n}")
