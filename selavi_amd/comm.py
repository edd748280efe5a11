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
    _disabled = None        # reason string once the preflight has switched the native path off for this process

    def __init__(self, handle, rank, world, group):
        self._h, self.rank, self.world, self.group = handle, rank, world, group

    @property
    def h(self):
        """The library's communicator handle; raises once a failed preflight has aborted (= freed) it, so that a holder of
        this object from before the failure (a sync_pair, an SK solver) cannot hand a dangling pointer to the C ABI."""
        if self._h is None:
            raise RuntimeError("selavi_amd.comm: this communicator was aborted by a failed preflight "
                               f"({NativeComm._disabled}); re-create the exchange (comm.sync_pair) on torch.distributed")
        return self._h

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
        if mode == "0" or cls._disabled or (mode != "force" and dist.get_backend(group) != "nccl"):
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

    @classmethod
    def preflight(cls, group=None, timeout_s=None, rounds=3):
        """Watchdog around the FIRST collectives of every communicator created so far: the step drives
        them CONCURRENTLY -- "bn" from the main stream, "bn_audio" from the audio trunk's stream, "grad" (100 MB buckets)
        from the buckets' stream -- which RCCL documents as safe only while all of them can be co-resident.  This issues
        exactly that pattern (`rounds` times: a 2C-double sum on bn and on bn_audio, a 100 MB fp32 mean on grad, each on a
        stream of its own), polls the streams' events against a deadline (SELAVI_COMM_TIMEOUT_S, default 120 s) and lets
        the ranks agree on the outcome over torch.distributed (MIN).  If ANY rank timed out or saw an error, every rank
        aborts its native communicators (ncclCommAbort), prints a LOUD warning and disables the native path for the
        process: callers re-create their exchanges on torch.distributed (SELAVI_NATIVE_COMM=0 behaviour).
        Returns True if the native path stays on (or was not in use)."""
        import sys
        import time
        import torch.distributed as dist
        gid_ = 0 if (group is None or group is dist.group.WORLD) else id(group)
        comms = {k: v for k, v in cls._cache.items() if v.world > 1 and k[0] == gid_}     # THIS group's communicators only:
        if not comms:                                                                   # the verdict below is agreed over it
            return True
        if timeout_s is None:
            timeout_s = float(os.environ.get("SELAVI_COMM_TIMEOUT_S", "120"))
        dev = torch.device("cuda", torch.cuda.current_device())
        main = torch.cuda.current_stream(dev)
        ok, why, events, keep = 1, "", [], []
        try:
            for (gid, tag), comm in comms.items():
                st = torch.cuda.Stream(device=dev)
                st.wait_stream(main)
                n = 25_000_000 if tag == "grad" else 2 * 512
                buf = torch.ones(n, dtype=torch.float32 if tag == "grad" else torch.float64, device=dev)
                keep.append((buf, st))
                with torch.cuda.stream(st):
                    for _ in range(rounds):
                        if tag == "grad":
                            comm.allreduce_avg_f32_(buf)
                        else:
                            comm.allreduce_(buf)
                            buf.mul_(1.0 / comm.world)
                    events.append((tag, comm, buf, st.record_event()))
            deadline = time.time() + timeout_s
            for tag, comm, buf, ev in events:
                while not ev.query():
                    if time.time() > deadline:
                        ok, why = 0, f"communicator {tag!r}: collective still running after {timeout_s:.0f} s"
                        break
                    C.slv_comm_async_error(comm.h)          # raises on an asynchronous RCCL error
                    time.sleep(0.002)
                if not ok:
                    break
                if abs(float(buf[0]) - 1.0) > 1e-6:          # ones stay ones under sum / world and under mean
                    ok, why = 0, f"communicator {tag!r}: wrong result {float(buf[0])!r}"
                    break
        except Exception as e:                                # SelaviHipError from a failing collective
            ok, why = 0, repr(e)
        if os.environ.get("SELAVI_COMM_PREFLIGHT_INJECT") == f"rank{dist.get_rank()}":     # test hook: this rank "fails"
            ok, why = 0, "injected failure (SELAVI_COMM_PREFLIGHT_INJECT)"
        flag = torch.tensor([ok], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            flag = flag.to(dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            return True
        sys.stderr.write("\n" + "!" * 100 + "\nselavi_amd.comm: the native RCCL communicators FAILED their preflight on rank %d (%s);\n"
                         "falling back to torch.distributed for SyncBN / Sinkhorn-Knopp / gradient buckets on EVERY rank "
                         "(SELAVI_NATIVE_COMM=0 behaviour).\n" % (dist.get_rank(), why or "another rank reported the failure")
                         + "!" * 100 + "\n")
        sys.stderr.flush()
        for key, comm in comms.items():
            try:
                C.slv_comm_abort(comm.h)
            except Exception:
                pass
            comm._h = None             # slv_comm_abort frees the handle: a holder of this object (a sync_pair made before
            cls._cache.pop(key, None)  # the failure) raises in allreduce_ instead of touching freed memory
        cls._disabled = why or "preflight failed on another rank"
        return False

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


# ---------------------------------------------------------------------------------------------------------------- diagnostics
DIAG = None          # {"syncbn": [(event, event)], "grad_wait": [...], "grad_bytes": n, "grad_collectives": n} while comm.diagnostics() is open


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()


class _Span:
    """HIP events on the current stream around an exchange (two records; measurement only)."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        if DIAG is not None:
            DIAG[self.kind].append((self.e0, e1))
        return False


def span(kind):
    """``with comm.span("syncbn"): <partials -> sums, all-reduce, sums -> coefficients>`` -- a no-op unless a
    ``comm.diagnostics()`` block is open (bench.py's N > 1 line)."""
    return _NOSPAN if DIAG is None else _Span(kind)


class diagnostics:
    """``with comm.diagnostics() as d: step(); step()`` then ``d.report(steps=2)``: what the data-parallel exchanges of those
    steps cost ON THE DEVICE, from HIP events on the streams they run on --
      * syncbn: every SyncBN exchange (main.py:117-118; partials -> fp64 sums, K-vector all-reduce, sums -> coefficients),
        count per step and mean / total duration;
      * grad: the gradient buckets (main.py:156-160): collectives and bytes per step, and the EXPOSED wait -- how long the
        compute stream stood still in the final callback until the last bucket had landed (0 = fully overlapped).
    Measurement only: the events add two records per exchange."""

    def __enter__(self):
        global DIAG
        DIAG = self.state = {"syncbn": [], "grad_wait": [], "grad_bytes": 0, "grad_collectives": 0, "grad_wait_host_s": 0.0}
        return self

    def __exit__(self, *exc):
        global DIAG
        DIAG = None
        return False

    def report(self, steps):
        torch.cuda.synchronize()
        st = self.state
        sb = [a.elapsed_time(b) for a, b in st["syncbn"]]
        gw = [a.elapsed_time(b) for a, b in st["grad_wait"]]
        n = max(steps, 1)
        return {"syncbn": {"exchanges_per_step": len(sb) / n, "mean_ms": (sum(sb) / len(sb)) if sb else None,
                           "max_ms": max(sb) if sb else None, "total_ms_per_step": sum(sb) / n,
                           "how": "HIP events around each exchange on its compute stream (kernels of the exchange included)"},
                "grad_allreduce": {"collectives_per_step": st["grad_collectives"] / n, "bytes_per_step": st["grad_bytes"] / n,
                                   "exposed_wait_ms_per_step": sum(gw) / n,
                                   "host_blocked_ms_per_step": st["grad_wait_host_s"] * 1e3 / n,
                                   "how": "HIP events around the compute stream's wait for the buckets' stream in the "
                                          "backward's final callback: 0 = the all-reduces were hidden behind backward"}}


def describe(group=None):
    """Which transport carries the exchanges of ``group`` in this process, and what RCCL itself reports about it."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"transport": "none (single process)"}
    gid_ = 0 if (group is None or group is dist.group.WORLD) else id(group)
    comms = {tag: c for (g, tag), c in NativeComm._cache.items() if g == gid_}
    out = {"world": dist.get_world_size(group), "bootstrap_backend": dist.get_backend(group),
           "native_comm_mode": NativeComm.mode(), "preflight": "failed: " + NativeComm._disabled if NativeComm._disabled else "ok"}
    if comms:
        any_c = next(iter(comms.values()))
        out["transport"] = "native: RCCL behind the C ABI (slv_comm_*) on the compute / bucket streams"
        out["library"] = any_c.library()
        # ncclCommCount per communicator: the ranks RCCL itself has in it (-1: the loaded library has no ncclCommCount)
        out["communicators"] = {tag: {"ranks_rccl_reports": int(C.slv_comm_count(c.h)), "world": c.world, "rank": c.rank}
                                for tag, c in sorted(comms.items())}
    else:
        out["transport"] = "torch.distributed (%s) collectives" % dist.get_backend(group)
        out["communicators"] = {}
    return out


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
