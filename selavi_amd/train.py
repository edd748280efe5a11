"""The training-step body of /root/reference/main.py:263-302 and the SK schedule of :163-171 as
reusable functions (main.py itself needs tensorboard/av/ffmpeg and is not importable)."""
import numpy as np
import torch

from .utils import get_loss


def train_step(model, optimizer, video, audio, selflabels, selected, headcount):
    """main.py:284-302.  Returns the detached loss tensor (no host sync here)."""
    feat_v, feat_a = model(video, audio)                         # :284
    if headcount == 1:
        labels = selflabels[selected, 0]                         # :287-288
    else:
        labels = selflabels[selected, :]                         # :289-290
    loss_vid = get_loss(feat_v, labels, headcount=headcount)     # :291
    loss_aud = get_loss(feat_a, labels, headcount=headcount)     # :292
    loss = 0.5 * loss_vid + 0.5 * loss_aud                       # :293
    optimizer.zero_grad()                                        # :296
    loss.backward()                                              # :301
    optimizer.step()                                             # :302
    return loss.detach()


def sk_schedule(epochs, n_batches, nopts=100, schedulepower=1.5):
    """main.py:168-170 (popped from the end: first SK at iteration 0)."""
    sched = (epochs * n_batches * (np.linspace(0, 1, nopts) ** schedulepower)[::-1]).tolist()
    return [(epochs + 2) * n_batches] + sched
