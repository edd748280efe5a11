"""CPU restatement of one SeLaVi training step (TEST INFRASTRUCTURE ONLY).

Follows ``/root/reference/main.py:284-302`` (forward, label gather, 0.5/0.5 loss, zero_grad,
backward, SGD step) with ``torch.optim.SGD(momentum=0.9, weight_decay=wd)`` exactly as
``main.py:132-137`` builds it.  Runs on torch's CPU kernels; used as the parity oracle for the
HIP engine and (bench.py ``cpu_baseline``) as the host-core baseline.
"""
import torch

from .model_ref import get_loss


def make_optimizer(model, lr=1e-2, wd=1e-5, momentum=0.9):
    return torch.optim.SGD(model.parameters(), lr=lr, momentum=momentum, weight_decay=wd)


def train_step(model, optimizer, video, audio, selflabels, selected, headcount):
    """One step; returns (loss, feat_v, feat_a).  ``model`` must be in train mode."""
    feat_v, feat_a = model(video, audio)                     # main.py:284
    if headcount == 1:
        labels = selflabels[selected, 0]                     # main.py:287-288
    else:
        labels = selflabels[selected, :]                     # main.py:289-290
    loss_vid = get_loss(feat_v, labels, headcount=headcount)  # main.py:291
    loss_aud = get_loss(feat_a, labels, headcount=headcount)  # main.py:292
    loss = 0.5 * loss_vid + 0.5 * loss_aud                   # main.py:293
    optimizer.zero_grad()                                    # main.py:296
    loss.backward()                                          # main.py:301
    optimizer.step()                                         # main.py:302
    return loss.detach(), feat_v, feat_a


def set_dropout_p(model, p):
    """Parity tests run with dropout disabled (p=0) or with an injected mask: the torch-CPU
    RNG stream cannot be reproduced on the device (SURVEY 7 'hard parts')."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p
