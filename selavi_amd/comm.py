"""RCCL communicator behind the C ABI (include/selavi_hip.h: slv_comm_*) for the hot path's exchanges.

The reference runs every collective through torch.distributed's NCCL process group (utils.py:133-146).  Here
torch.distributed only bootstraps: it carries the 128-byte RCCL unique id from rank 0 to the other ranks once; after
that the SyncBN sums (main.py:117-118), the Sinkhorn-Knopp column sums and the small fp64 reductions of the SK round are
RCCL calls issued by libselavi_hip.so on the CALLER's stream, in order with the kernels around them -- no
process-group stream, no event hops, and the (partials -> sums, all-reduce, sums -> coefficients) sequence of a SyncBN
layer is one library call (slv_bn_sync_finalize).

Used when the process group's backend is "nccl" (= RCCL on ROCm) and SELAVI_NATIVE_COMM != 0; with any other backend
(the gloo test rigs that put several ranks on one GPU) the callers keep using torch.distributed -- unless
SELAVI_NATIVE_COMM=force, which takes the native path whatever carries the bootstrap (tests/test_native_comm_gpu.py runs
two ranks on one GPU that way, with SELAVI_RCCL_LIB naming the shared-memory test double of librccl).

One communicator per (group, tag): collectives on ONE RCCL communicator serialise in issue order whatever stream they
are on, so every concurrently running consumer has its own -- "bn" (video trunk + heads, main stream), "bn_audio" (the
audio trunk's side stream), "grad" (the gradient buckets' stream, parallel.GradSink).  Creation is collective and happens
in program order (audio, video, heads, buckets), the same on every rank.
"""
import ctypes
import os

import torch

from ._lib import C, ptr, stream


class NativeComm:
    _cache = {}

    def __init__(self, handle, rank, world, group):
        self.h, self.rank, self.world, self.group = handle, rank, world, group

    @staticmethod
    def mode():
        """SELAVI_NATIVE_COMM: "0" never, "1" (default) when the group's backend is nccl, "force" always."""
        return os.environ.get("SELAVI_NATIVE_COMM", "1")

    @classmethod
    def for_group(cls, group=None, tag="bn"):
        """The communicator ``tag`` mirroring ``group`` (None: the default group), created on first use -- COLLECTIVE
        over the group then, cached afterwards.  None when the native path does not apply (see module docstring)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        mode = cls.mode()
        if mode == "0" or (mode != "force" and dist.get_backend(group) != "nccl"):
            return None
        key = (0 if (group is None or group is dist.group.WORLD) else id(group), tag)   # WORLD and None are the same group
        got = cls._cache.get(key)
        if got is not None:
            return got
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lib = os.environ.get("SELAVI_RCCL_LIB")
        if not lib:
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")     # one RCCL runtime per process
            lib = bundled if os.path.exists(bundled) else None
        C.slv_comm_load(lib.encode() if lib else None)
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            C.slv_comm_unique_id(idbuf.data_ptr())
        src = dist.get_global_rank(group, 0) if (group is not None and group is not dist.group.WORLD) else 0
        if dist.get_backend(group) == "nccl":
            t = idbuf.to(torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(t, src=src, group=group)
            idbuf = t.cpu()
        else:
            dist.broadcast(idbuf, src=src, group=group)
        handle = ctypes.c_void_p()
        C.slv_comm_init(ctypes.addressof(handle), idbuf.data_ptr(), rank, world)
        got = cls._cache[key] = cls(handle, rank, world, group)
        return got

    @classmethod
    def destroy_all(cls):
        """Destroy every cached communicator (collective in RCCL: every rank calls it at the same point)."""
        for comm in cls._cache.values():
            C.slv_comm_destroy(comm.h)
        cls._cache.clear()

    def allreduce_(self, t):
        """In-place sum on the current stream (fp64 / fp32 / int64 tensors)."""
        assert t.is_cuda and t.is_contiguous()
        if t.dtype == torch.float64:
            C.slv_comm_allreduce_f64(self.h, ptr(t), t.numel(), stream())
        elif t.dtype == torch.float32:
            C.slv_comm_allreduce_f32(self.h, ptr(t), t.numel(), 0, stream())
        elif t.dtype == torch.int64:
            C.slv_comm_allreduce_i64(self.h, ptr(t), t.numel(), stream())
        else:
            raise TypeError(t.dtype)
        return t

    def allreduce_avg_f32_(self, t):
        """In-place MEAN over the ranks on the current stream (ncclAvg): the gradient buckets."""
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        C.slv_comm_allreduce_f32(self.h, ptr(t), t.numel(), 1, stream())
        return t

    def library(self):
        p = C.slv_comm_library()
        return p.decode() if p else ""


def allreduce_sum_(t, where):
    """Sum ``t`` in place over ``where``: a NativeComm (RCCL call on the current stream) or a torch process group."""
    if isinstance(where, NativeComm):
        return where.allreduce_(t)
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=where)
    return t


def sync_pair(group=None, tag="bn"):
    """(where, world) for SyncBN over ``group``: the native communicator ``tag`` when it applies, else the torch group."""
    import torch.distributed as dist
    comm = NativeComm.for_group(group, tag)
    return (comm if comm is not None else group, dist.get_world_size(group))
