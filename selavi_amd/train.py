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
    if kind != "native":
        raise ValueError(f"unknown data-parallel kind {kind!r} (native | ddp)")
    from .parallel import DataParallel
    return DataParallel(model, group=group)      # errors propagate: no silent change of the wrapper


def _join_package_streams():
    """The current stream waits for every side stream this package created (audio trunk, weight gradients)."""
    import torch
    from . import engine, model
    cur = torch.cuda.current_stream()
    for st in list(model._SIDE_STREAMS.values()) + list(engine._WGRAD_STREAMS.values()):
        if st.device == cur.device:
            cur.wait_stream(st)


class GraphedStep:
    """The whole training step (forward, loss, backward, SGD) captured once into a HIP graph and replayed.

    For shapes where the ~700 kernel launches of a step cost more host time than GPU time (BASELINE configs[0]: bs 4,
    8 frames) the step is bound by Python/ctypes enqueue; a graph replay is one launch.  Every kernel of this package
    is capturable as is (no allocation, synchronisation or host copy inside libselavi_hip.so; pointer tables travel in
    kernel arguments; the trunk / weight-gradient side streams fork from and join the capturing stream with events).
    Inputs are static buffers: copy the next batch into ``video / audio / selected`` (and update ``selflabels`` in
    place) before ``replay()``.  BatchNorm's ``num_batches_tracked`` advances per replay.
    Data parallel: ``model`` may be a ``parallel.DataParallel`` whose exchanges run on the library's own RCCL communicators
    (comm.NativeComm: SyncBN = slv_bn_sync_finalize on the compute streams, the gradient buckets = slv_comm_allreduce_f32 on
    their own stream, forked from and joined to the capturing stream with events) -- the collectives are graph nodes like
    the kernels around them, every rank replays the same graph.  The per-step equal-batch check (a host read) moves from
    ``forward`` to ``replay()``.  A wrapper on torch.distributed collectives (gloo, torch DDP) is refused."""

    def __init__(self, model, optimizer, video, audio, selflabels, selected, headcount, warmup=3):
        import torch
        self.model, self.video, self.audio, self.selflabels, self.selected = model, video, audio, selflabels, selected
        self._dp = model if hasattr(model, "check_equal_batches") else None
        if self._dp is not None and not self._dp.capturable():
            raise RuntimeError("train.GraphedStep: this DataParallel's exchanges go through torch.distributed (host-driven): "
                               "only the library's RCCL communicators (backend nccl / SELAVI_NATIVE_COMM) can be captured")
        if self._dp is None and type(model).__name__ == "DistributedDataParallel":
            raise RuntimeError("train.GraphedStep: torch DDP's bucket hooks are not capturable; use train.data_parallel(kind='native')")
        self._bns = [m for m in model.modules() if hasattr(m, "note_batch")]
        # (the audio trunk keeps its own stream under capture: its node forks and joins with events, nn.TrunkFunction)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # plans, tuning, momentum buffers, side streams: all before capture
            for _ in range(warmup):
                train_step(model, optimizer, video, audio, selflabels, selected, headcount)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import nn as snn
        snn.dropout_device_state(video.device, create=True)     # Dropout(0.3) draws fresh masks on every replay
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # (under torch.distributed's nccl backend a watchdog thread polls its work events with hipEventQuery: legal beside a
        #  capture only in thread-local error mode -- the default "global" mode makes that query fail and takes the process down)
        mode = {"capture_error_mode": "thread_local"} if self._dp is not None else {}
        with torch.cuda.graph(self.graph, **mode):
            self.loss = train_step(model, optimizer, video, audio, selflabels, selected, headcount)
            _join_package_streams()                    # a capture may only end with every forked stream joined
        for b in self._bns:                            # the captured call counted one batch without running it
            b._pending -= 1

    def replay(self):
        if self._dp is not None:
            self._dp.check_equal_batches(int(self.video.shape[0]), self.video.device)
        self.graph.replay()
        for b in self._bns:
            b.note_batch()
        return self.loss
