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


_SIDE_GROUPS = {}


def _side_group(backend):
    """The second process group the default group's buckets travel on (one per process, created collectively once)."""
    g = _SIDE_GROUPS.get(backend)
    if g is None:
        g = _SIDE_GROUPS[backend] = dist.new_group(ranks=list(range(dist.get_world_size())), backend=backend)
    return g


class GradSink:
    """Flat gradient buffers per autograd node + their asynchronous all-reduce."""

    def __init__(self, group=None, own_group=None):
        """The buckets travel on a communicator and a stream of their OWN: collectives on one communicator serialise,
        and the SyncBN exchanges of the layers still running backward (tiny, on the critical path) would otherwise
        queue behind layer4's 100 MB bucket.

        * native path (comm.NativeComm applies: backend nccl, or SELAVI_NATIVE_COMM=force): the library's own RCCL
          communicator "grad" (``slv_comm_allreduce_f32(average=1)``) on a dedicated HIP stream -- together with the
          SyncBN communicators ONE RCCL runtime serves the step, torch.distributed only bootstraps;
        * otherwise a torch process group: for the default group a second one is created (``own_group``, default on,
          SELAVI_DP_OWN_GROUP=0 switches it off; ``dist.new_group`` is collective over ALL ranks of the default group,
          so it is only done there), cached per rank set; a caller that passes a sub-group passes the group the
          buckets shall use."""
        import os
        from .comm import NativeComm
        if own_group is None:
            own_group = os.environ.get("SELAVI_DP_OWN_GROUP", "1") == "1"
        backend = dist.get_backend(group)
        self.native = NativeComm.for_group(group, "grad")
        self.stream = None
        if self.native is None and own_group and dist.get_world_size(group) > 1 and (group is None or group is dist.group.WORLD):
            group = _side_group(backend)
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
        from . import comm as _comm
        if _comm.DIAG is not None:                # bench.py's N > 1 line: collectives / bytes per step
            _comm.DIAG["grad_collectives"] += len(tensors)
            _comm.DIAG["grad_bytes"] += sum(t.numel() * t.element_size() for t in tensors)
        if self.native is not None:               # RCCL behind the C ABI, on the buckets' own stream
            cur = torch.cuda.current_stream()
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(cur)          # the node's kernels have written the views
            with torch.cuda.stream(self.stream):
                for t in tensors:
                    self.native.allreduce_avg_f32_(t)
                    t.record_stream(self.stream)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.pending.append((ev, None))
            torch.autograd.Variable._execution_engine.queue_callback(self.finalize)
            return
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        for t in tensors:
            self.pending.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))
        # we are inside backward: hold its caller until the buffers have landed.  One callback per node (the first to run
        # waits for everything launched so far, the others find nothing): no state survives an interrupted backward
        torch.autograd.Variable._execution_engine.queue_callback(self.finalize)

    def finalize(self):
        pending, self.pending = self.pending, []
        if not pending:
            return
        from . import comm as _comm
        if _comm.DIAG is None:
            return self._wait(pending)
        import time
        t0 = time.perf_counter()
        with _comm.span("grad_wait"):             # (diagnostics only) how long the compute stream stands still here
            self._wait(pending)
        _comm.DIAG["grad_wait_host_s"] += time.perf_counter() - t0      # (host-driven transports block here instead)

    def _wait(self, pending):
        for work, t in pending:
            if t is None:                         # native path: an event on the buckets' stream
                torch.cuda.current_stream().wait_event(work)
                continue
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
        from .comm import NativeComm
        if not NativeComm.preflight(group):       # loud fall-back: the exchanges are re-created on torch.distributed
            module.set_sync_bn(True, group)
            self.sink = GradSink(group)
            assert self.sink.native is None
        module.set_grad_sink(self.sink)
        self.group = group
        self._last_batch = None
        self._batch_checks = []        # [(event | None, tensor [max b, max -b], b of this rank)] not yet read

    def _read_batch_checks(self, block):
        while self._batch_checks:
            ev, t, b = self._batch_checks[0]
            if ev is not None:
                if block:
                    ev.synchronize()
                elif not ev.query():
                    return
            self._batch_checks.pop(0)
            hi, lo = int(t[0]), -int(t[1])
            if hi != lo:
                raise RuntimeError(f"selavi_amd.DataParallel: per-rank batch sizes differ (this rank {b}, max {hi}, "
                                   f"min {lo}); SyncBN / the averaged buckets assume equal batches")

    def check_equal_batches(self, b, device):
        """slv_bn_sync_finalize and the mean all-reduce of the buckets assume EQUAL per-rank batches (the reference's loader
        drops the ragged last batch, main.py:95-103).  The check is a collective, so EVERY rank issues it on EVERY call (a
        rank-local condition would leave the one rank with the ragged batch alone in the all-reduce: a hang on the native
        communicators, a mismatched pairing on torch.distributed).  It never stalls the host in the steady state: on RCCL
        the result lands in pinned memory behind an event.  The result of the PREVIOUS call is read -- blocking on its event,
        which completed at the head of the previous step -- BEFORE this call launches anything: a ragged batch on another rank
        raises here on every rank, one step late at most and before this step's SyncBN / bucket collectives could wait for a
        rank that has already left; a rank whose OWN batch size just changed (the first call, its ragged batch) also waits
        for this call's result before launching the step."""
        self._read_batch_checks(block=True)                                # the previous call's verdict (steady state: ready)
        if dist.get_backend(self.group) == "nccl":
            t = torch.full((2,), b, dtype=torch.int64, device=device)
            t[1] = -b                                                  # (fill kernels: no host-to-device copy, no sync)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)      # stream-ordered, asynchronous to the host
            host = torch.empty(2, dtype=torch.int64, pin_memory=True)
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._batch_checks.append((ev, host, b))
        else:
            t = torch.tensor([b, -b], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._batch_checks.append((None, t, b))
        if b != self._last_batch:
            self._read_batch_checks(block=True)
        self._last_batch = b

    def forward(self, *args, **kwargs):
        # (under HIP-graph capture -- train.GraphedStep -- the check is issued by replay(): it reads a result on the host)
        if args and hasattr(args[0], "shape") and not torch.cuda.is_current_stream_capturing():
            self.check_equal_batches(int(args[0].shape[0]), args[0].device)
        return self.module(*args, **kwargs)

    def capturable(self):
        """Can a step through this wrapper be captured into a HIP graph?  Only when EVERY exchange of the step is a library
        call on a HIP stream (comm.NativeComm: the RCCL launches are captured like any kernel); torch.distributed's
        collectives (gloo, or the fall-back after a failed preflight) are host-driven."""
        from .comm import NativeComm
        return self.sink.native is not None and NativeComm._disabled is None
