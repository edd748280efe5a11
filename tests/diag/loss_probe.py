import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_train16_gpu import _step_setup
from oracle.model_ref import portable_fill_
from selavi_amd.utils import get_loss
B, T, S, hc, K = 8, 8, 64, 2, 7
for prec in ("bf16", "fp32"):
    m, opt, video, audio, sl, sel, _ = _step_setup(prec, hc=hc, K=K, B=4, T=T, S=S)
    video = portable_fill_(torch.empty(B, 3, T, S, S), 5).cuda()
    audio = portable_fill_(torch.empty(B, 1, 40, 36), 6).cuda()
    sel = torch.tensor([3, 17, 42, 63, 5, 9, 33, 60])[:B].cuda()
    if prec == "fp32":
        with torch.no_grad():
            for p in m.video_network.parameters():
                if p.dim() == 5:
                    p.copy_(p.to(torch.bfloat16).float())
        video = video.to(torch.bfloat16).float()
    fv, fa = m(video, audio)
    labels = sl[sel, :]
    lv, la = get_loss(fv, labels, headcount=hc), get_loss(fa, labels, headcount=hc)
    print(prec, "feat_v norm %.6f sum %.6f" % (float(torch.stack(list(fv)).float().norm()), float(torch.stack(list(fv)).float().sum())))
    print(prec, "loss_v %.6f loss_a %.6f total %.6f" % (float(lv), float(la), float(0.5 * lv + 0.5 * la)))
