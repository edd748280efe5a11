"""Data parallelism for the SeLaVi step without per-parameter work (one process per GPU, RCCL over xGMI).

The reference wraps the model in ``torch.nn.parallel.DistributedDataParallel`` (main.py:156-160).  DDP works with this
package (``train.wrap_ddp``), but it pays one scaling/copy kernel per parameter and per step (167 parameters:
+2.2 ms on the 44.8 ms step, tools/dist_overhead.py) to move gradients into its buckets.  Here the buckets ARE the
gradients: every autograd node of the model (heads; video layer4 .. stem; audio trunk -- the order backward reaches
them) owns one flat fp32 buffer, the weight-gradient / BatchNorm-backward kernels write straight into views of it,
``param.grad`` is that view, and the node ends by launching ONE asynchronous all-reduce (average) of its buffer on
RCCL's stream while backward continues -- 7 collectives per step (178 MB), no per-parameter kernels.  The step's
optimizer is held back until the collectives have landed by an autograd final callback.

``DataParallel(model)`` has DDP's surface as far as the reference uses it: ``.module``, ``forward``, ``state_dict``
keys prefixed ``module.``; SyncBN is switched on (main.py:117-118) and parameters/buffers are broadcast from rank 0
once.  Gradients are overwritten, not accumulated, by each backward (the reference never accumulates).
"""
import torch
import torch.distributed as dist

_ALIGN = 64          # floats: every gradient view starts on a 256-byte boundary


class GradSink:
    """Flat gradient buffers per autograd node + their asynchronous all-reduce."""

    def __init__(self, group=None, own_group=None):
        """own_group (default on, SELAVI_DP_OWN_GROUP=0 switches it off): the buckets travel on a process group of
        their own -- a second RCCL communicator and stream.  Collectives on one communicator serialise, and the
        SyncBN exchanges of the layers still running backward (tiny, on the critical path) would otherwise queue
        behind layer4's 100 MB bucket."""
        import os
        if own_group is None:
            own_group = os.environ.get("SELAVI_DP_OWN_GROUP", "1") == "1"
        backend = dist.get_backend(group)
        if own_group and dist.get_world_size(group) > 1:
            ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
            group = dist.new_group(ranks=ranks, backend=backend)       # collective over the default group's ranks
        self.group = group
        self.world = dist.get_world_size(group)
        self.avg = backend == "nccl"              # RCCL averages in the collective; gloo: sum, then one scale per buffer
        self.flat = {}                            # key -> (flat buffer, {id(param): view})
        self.pending = []

    def views(self, key, params):
        ent = self.flat.get(key)
        if ent is None:
            offs, n = [], 0
            for p in params:
                offs.append(n)
                n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
            ent = self.flat[key] = (flat, {id(p): flat[o:o + p.numel()].view_as(p) for p, o in zip(params, offs)})
        return ent[1]

    def carve(self, key, shapes):
        """One flat buffer for a group of tensors the caller's kernels will write (the heads' grouped gradients)."""
        ent = self.flat.get(key)
        if ent is None:
            offs, n = [], 0
            for shp in shapes:
                offs.append(n)
                cnt = 1
                for d in shp:
                    cnt *= d
                n += (cnt + _ALIGN - 1) // _ALIGN * _ALIGN
            flat = torch.zeros(n, dtype=torch.float32, device=torch.cuda.current_device())
            views = []
            for shp, o in zip(shapes, offs):
                cnt = 1
                for d in shp:
                    cnt *= d
                views.append(flat[o:o + cnt].view(*shp))
            ent = self.flat[key] = (flat, views)
        return ent

    def deliver(self, key, params):
        """The node's kernels have written every view: publish them as .grad and start the node's all-reduce."""
        flat, views = self.flat[key]
        for p in params:
            p.grad = views[id(p)]
        self.reduce([flat])

    def reduce(self, tensors):
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        for t in tensors:
            self.pending.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))
        # we are inside backward: hold its caller until the buffers have landed.  One callback per node (the first to run
        # waits for everything launched so far, the others find nothing): no state survives an interrupted backward
        torch.autograd.Variable._execution_engine.queue_callback(self.finalize)

    def finalize(self):
        pending, self.pending = self.pending, []
        for work, t in pending:
            work.wait()
            if not self.avg:
                t.mul_(1.0 / self.world)


class DataParallel(torch.nn.Module):
    def __init__(self, module, group=None, broadcast=True):
        super().__init__()
        self.module = module
        if broadcast:                             # like DDP's constructor: every rank starts from rank 0's state
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for t in module.state_dict().values():
                dist.broadcast(t, src=src, group=group)
        module.set_sync_bn(True, group)
        self.sink = GradSink(group)
        module.set_grad_sink(self.sink)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
