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


def wrap_ddp(model, device_ids, **kw):
    """DistributedDataParallel (main.py:156-160) with the settings this model allows -- measured on one MI355X with
    every collective going through RCCL on a world of one rank (tools/dist_overhead.py): default DDP costs 4.8 ms on
    the 44.8 ms SyncBN step, this configuration 2.2 ms.

    * broadcast_buffers=False: the default re-broadcasts ~200 BatchNorm buffers before every forward.  Under SyncBN
      every rank finalises the same all-reduced sums with the same kernel, so the running statistics are
      bit-identical on all ranks by construction (tests/test_cluster_gpu.py asserts it) -- nothing to broadcast.
    * gradient_as_bucket_view=True: .grad tensors live inside the all-reduce buckets (no grad <-> bucket copies).

    static_graph=True would save another 1.8 ms but is NOT used: in the two-rank test the ranks' conv weights then
    diverge from the first step on (DDP's first-iteration bookkeeping leaves those gradients un-reduced;
    tests/diag/ddp_lockstep.py bisects the options).
    """
    import torch
    opts = dict(broadcast_buffers=False, gradient_as_bucket_view=True)
    opts.update(kw)
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, **opts)


def data_parallel(model, device_ids, group=None, kind=None):
    """The data-parallel wrapper of a training run: ``parallel.DataParallel`` (flat per-node gradient buckets written
    by the kernels, 7 all-reduces per step) unless ``kind``/SELAVI_DP says "ddp" (torch DDP through wrap_ddp)."""
    import os
    kind = kind or os.environ.get("SELAVI_DP", "native")
    if kind == "ddp":
        return wrap_ddp(model, device_ids, process_group=group)
    from .parallel import DataParallel
    try:
        return DataParallel(model, group=group)
    except (AttributeError, TypeError, NotImplementedError) as e:      # a torch build without an API used there:
        import warnings                                                 # deterministic, so every rank falls back alike
        warnings.warn(f"selavi_amd.parallel.DataParallel unavailable ({e!r}); using torch DDP")
        model.set_grad_sink(None)
        return wrap_ddp(model, device_ids, process_group=group)
