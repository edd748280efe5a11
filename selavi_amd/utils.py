"""Mirror of the hot-path helpers of /root/reference/utils.py (get_loss :377-387,
warmup_batchnorm :389-418) on the HIP kernels."""
import time

import torch

from . import nn as snn


def get_loss(activations, targets, headcount=1):
    """utils.py:377-387: cross entropy (hc==1) or the mean over heads of per-head cross entropy.

    ``activations``: B x K tensor (hc==1) or the list of hc B x K tensors AVModel returns;
    ``targets``: int64 [B] or [B, hc].  One grouped softmax-CE kernel for all heads."""
    if headcount == 1:
        act = activations[0] if isinstance(activations, (list, tuple)) else activations
        return snn.GroupedCE.apply(act.unsqueeze(0), targets.reshape(-1, 1))
    stacked = getattr(activations, "stacked", None)
    if stacked is None:
        stacked = torch.stack(list(activations))
    return snn.GroupedCE.apply(stacked, targets)


def warmup_batchnorm(args, model, dataloader, batches=20, group=None):
    """utils.py:389-418: `batches` no-grad train-mode forwards to seed the BN running statistics.
    (The reference reads an undefined ``args.distributed`` at :412; here the barrier is taken
    whenever torch.distributed is initialised.)"""
    start = time.time()
    with torch.no_grad():
        model.train()
        for i, batch in enumerate(dataloader):
            video, audio = batch[0], batch[1]
            if i == batches:
                break
            _ = model(video.cuda(non_blocking=True), audio.cuda(non_blocking=True))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier(group=group) if group is not None else dist.barrier()
    return time.time() - start
