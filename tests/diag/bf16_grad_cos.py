"""Per-parameter agreement of the bf16-path gradients with the fp32 HIP step (diagnostic, not a test).
    python tests/diag/bf16_grad_cos.py B T S [damp]
Three runs from one initialisation: fp32; fp32 with conv weights and the clip rounded to bf16 (how much of the
disagreement is the network's own sensitivity at random init); the bf16 path.  damp: the last BatchNorm of every
residual block starts at gamma = damp (blocks close to the identity: a well-conditioned backward)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_train16_gpu import _step_setup
from selavi_amd.utils import get_loss

B, T, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
damp = float(sys.argv[4]) if len(sys.argv) > 4 else None
res = {}
for prec in ("fp32", "fp32r", "bf16"):
    m, opt, video, audio, sl, sel, hc = _step_setup("bf16" if prec == "bf16" else "fp32", B=B, T=T, S=S)
    with torch.no_grad():
        if damp is not None:
            for li in range(1, 5):
                for blk in getattr(m.video_network.base, f"layer{li}"):
                    blk.conv2[1].weight.fill_(damp)
        if prec == "fp32r":
            for n, p in m.video_network.named_parameters():
                if p.dim() == 5:
                    p.copy_(p.to(torch.bfloat16).float())
            video = video.to(torch.bfloat16).float()
    fv, fa = m(video, audio)
    labels = sl[sel, :]
    loss = 0.5 * get_loss(fv, labels, headcount=hc) + 0.5 * get_loss(fa, labels, headcount=hc)
    opt.zero_grad()
    loss.backward()
    res[prec] = (float(loss.detach()), {n: p.grad.detach().double().flatten() for n, p in m.named_parameters() if n.startswith("video")})
print("loss", {k: v[0] for k, v in res.items()})
rows = []
for n, a in res["fp32"][1].items():
    out = []
    for other in ("fp32r", "bf16"):
        b = res[other][1][n]
        out.append((float((a @ b) / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (a.norm() + 1e-30))))
    rows.append((out[1][0], out, float(a.norm()), a.numel(), n))
for cosb, out, nrm, cnt, n in sorted(rows):
    print(f"bf16 cos {out[1][0]:7.4f} rel {out[1][1]:7.4f} | fp32-rounded cos {out[0][0]:7.4f} rel {out[0][1]:7.4f} | |g| {nrm:9.3e} {cnt:8d} {n}")
