"""Sinkhorn-Knopp pseudo-label solver -- host side (mirror of the reference's src/sk_utils.py).

``optimize_L_sk_gpu(args, PS, hc, logger)`` keeps the reference signature and semantics
(/root/reference/src/sk_utils.py:359-422): it reads ``args.distribution, args.dist,
args.diff_dist_every, args.diff_dist_per_head, args.gauss_sd, args.headcount, args.lamb,
args.rank``, writes ``args.dist``, destroys ``PS`` (raised to lamb/2 in place) and returns
``(cost: float, newL: int64 tensor on the device)``.  ``optimize_L_sk`` is an alias (the name
BASELINE.json's north star uses).

All arithmetic on the N x K matrix runs in libselavi_hip.so (csrc/sk.hip): one fused HBM pass per
iteration, device-side loop control (the host only polls a 32-byte status word every 10
iterations, two batches in flight), fixed-order reductions.  With ``group`` given the rows are
sharded over the ranks of that process group: one all-reduce of K+1 fp64 per
iteration over RCCL, labels stay local to the shard.
"""
import math
import time

import torch

from ._lib import C, ptr, stream


class HipSkBackend:
    """Thin object wrapper over the slv_sk_* C ABI (the only backend the product ships).

    tests/ injects a numpy test double with the same methods to exercise the multi-rank
    control flow over gloo on CPU; nothing in this package falls back to it."""

    name = "hip"

    def device_of(self, t):
        return t.device

    def workspace(self, K, grid, device):
        nbytes = C.slv_sk_workspace_bytes(K, grid)
        return torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=device)

    def default_grid(self, N, K):
        return C.slv_sk_default_grid(N, K)

    def s_view(self, ws, K, grid):
        off = (C.slv_sk_s_ptr(ptr(ws), K, grid) - ws.data_ptr()) // 8
        return ws[off:off + K + 1]

    def alpha_view(self, ws, K, grid):
        off = (C.slv_sk_alpha_ptr(ptr(ws), K, grid) - ws.data_ptr()) // 8
        return ws[off:off + K]

    def pow_(self, P, power):
        C.slv_sk_pow(ptr(P), P.numel(), float(power), stream())

    def colsum(self, P, weight, ws, grid):
        N, K = P.shape
        out = torch.empty(K, dtype=torch.float64, device=P.device)
        C.slv_sk_colsum(ptr(P), ptr(weight), N, K, ptr(out), ptr(ws), grid, stream())
        return out

    def begin(self, P, N_global, beta, ws, grid):
        N, K = P.shape
        C.slv_sk_begin(ptr(P), N, N_global, K, ptr(beta), ptr(ws), grid, stream())

    def pass_(self, P, N_global, beta, ws, grid):
        N, K = P.shape
        C.slv_sk_pass(ptr(P), N, N_global, K, ptr(beta), ptr(ws), grid, stream())

    def local_reduce(self, K, ws, grid):
        C.slv_sk_local_reduce(K, ptr(ws), grid, stream())

    def update(self, r, K, tol, max_iter, first, ws, grid):
        C.slv_sk_update(ptr(r), K, float(tol), int(max_iter), int(first), ptr(ws), grid, stream())

    def iterate(self, P, beta, r, tol, max_iter, n_iters, ws, grid):
        N, K = P.shape
        C.slv_sk_iterate(ptr(P), N, K, ptr(beta), ptr(r), float(tol), int(max_iter), int(n_iters),
                         ptr(ws), grid, stream())

    def status_async(self, ws, K, grid, host_buf):
        C.slv_sk_status(ptr(ws), K, grid, host_buf.data_ptr(), stream())
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def labels(self, P, beta, ws, grid):
        N, K = P.shape
        L = torch.empty(N, dtype=torch.int64, device=P.device)
        logsum = torch.empty(1, dtype=torch.float64, device=P.device)
        C.slv_sk_labels(ptr(P), N, K, ptr(beta), ptr(ws), grid, ptr(L), ptr(logsum), stream())
        return L, logsum

    def host_status_buffer(self):
        return torch.zeros(4, dtype=torch.float64).pin_memory()


_HIP = HipSkBackend()


def head_probabilities(logits_v, logits_a, power=1.0, out=None):
    """PS = softmax64(logits_v) * softmax64(logits_a) [** power]  in one fused kernel.

    Mirrors sk_utils.py:309-315 (+ the ``PS.pow_`` of :391 when ``power != 1``)."""
    assert logits_v.shape == logits_a.shape and logits_v.dtype == torch.float32
    N, K = logits_v.shape
    P = out if out is not None else torch.empty(N, K, dtype=torch.float64, device=logits_v.device)
    C.slv_sk_prepare(ptr(logits_v.contiguous()), ptr(logits_a.contiguous()), ptr(P), N, K, float(power),
                     stream())
    return P


def softmax64(logits):
    """``F.softmax(x, dim=1, dtype=torch.float64)`` (sk_utils.py:208-211) on the device."""
    N, K = logits.shape
    P = torch.empty(N, K, dtype=torch.float64, device=logits.device)
    C.slv_sk_softmax64(ptr(logits.contiguous()), ptr(P), N, K, stream())
    return P


def _allreduce(t, group):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def sinkhorn(P, r, lamb, N_global=None, group=None, backend=None, tol=1e-1, max_iter=2000,
             batch=10, grid=None, already_powered=False):
    """Run the SK loop on ``P`` (N_local x K fp64, destroyed).  Returns (labels, logsum, info).

    ``r``: K normalised marginals (sk_utils.py:392-393).  ``group``: torch.distributed process group
    whose ranks each hold a row shard (None = single GPU).  The loop follows sk_utils.py:400-406
    exactly: err is tested on counters 0,10,20,... only, so the iteration count is == 1 (mod 10)
    unless the 2000 cap hits."""
    be = backend or _HIP
    N, K = P.shape
    N_global = N if N_global is None else N_global
    dev = be.device_of(P)
    grid = grid or be.default_grid(N, K)
    ws = be.workspace(K, grid, dev)
    beta = torch.empty(N, dtype=torch.float64, device=dev)
    r = r.reshape(K).contiguous()
    if not already_powered:
        be.pow_(P, 0.5 * lamb)                                  # :391
    be.begin(P, N_global, beta, ws, grid)                       # beta = 1/N (:390), s0
    be.local_reduce(K, ws, grid)
    if group is not None:
        _allreduce(be.s_view(ws, K, grid), group)
    be.update(r, K, tol, max_iter, True, ws, grid)              # alpha0 = r / s0
    host = [be.host_status_buffer(), be.host_status_buffer()]
    pending = []
    n_enq = 0
    status = None
    while True:
        # keep two batches in flight; kernels after `done` are device-side no-ops
        while len(pending) < 2 and n_enq < max_iter + batch:
            if group is None:
                be.iterate(P, beta, r, tol, max_iter, batch, ws, grid)
            else:
                for _ in range(batch):
                    be.pass_(P, N_global, beta, ws, grid)
                    be.local_reduce(K, ws, grid)
                    _allreduce(be.s_view(ws, K, grid), group)  # K col sums + err in one message
                    be.update(r, K, tol, max_iter, False, ws, grid)
            n_enq += batch
            hb = host[(n_enq // batch) % 2]
            pending.append((be.status_async(ws, K, grid, hb), hb))
        ev, hb = pending.pop(0)
        ev.synchronize()
        status = hb.clone()
        if status[1] != 0 or not pending and n_enq >= max_iter + batch:
            break
    for ev, _ in pending:
        ev.synchronize()
    L, logsum = be.labels(P, beta, ws, grid)
    info = dict(iters=int(status[0]), err=float(status[2]), alpha=be.alpha_view(ws, K, grid).clone(),
                beta=beta, grid=grid)
    return L, logsum, info


def optimize_L_sk_gpu(args, PS, hc, logger=None, group=None, N_global=None, backend=None):
    """Drop-in for ``sk_utils.optimize_L_sk_gpu`` (sk_utils.py:359-422).  Extra keyword arguments
    (not in the reference) enable the row-sharded multi-GPU form."""
    be = backend or _HIP
    N, K = PS.shape
    Ng = N if N_global is None else N_global
    dev = be.device_of(PS)
    tt = time.time()
    _K_dist = torch.ones((K, 1), dtype=torch.float64, device=dev)                      # :366
    if args.distribution != 'default':
        grid = be.default_grid(N, K)
        colsum = be.colsum(PS, None, be.workspace(K, grid, dev), grid)                  # PS.sum(0) :368
        if group is not None:
            _allreduce(colsum, group)
        marginals_argsort = torch.argsort(colsum)
        if (args.dist is None) or args.diff_dist_every:
            if args.distribution == 'gauss':
                if args.diff_dist_per_head:
                    _K_dists = [(torch.randn(size=(K, 1), dtype=torch.float64, device=dev)
                                 * args.gauss_sd + 1) * Ng / K for _ in range(args.headcount)]  # :371-373
                    if group is not None:
                        import torch.distributed as dist
                        for d in _K_dists:
                            dist.broadcast(d, src=dist.get_global_rank(group, 0)
                                           if hasattr(dist, "get_global_rank") else 0, group=group)
                    args.dist = _K_dists
                    _K_dist = _K_dists[hc]
                else:
                    _K_dist = (torch.randn(size=(K, 1), dtype=torch.float64, device=dev)
                               * args.gauss_sd + 1) * Ng / K                          # :377
                    _K_dist = torch.clamp(_K_dist, min=1)                              # :378
                    if group is not None:
                        import torch.distributed as dist
                        dist.broadcast(_K_dist, src=0, group=group)
                    args.dist = _K_dist
            if getattr(args, "rank", 0) == 0 and logger is not None:
                logger.info(f"distribution used: {_K_dist}")
        else:
            _K_dist = args.dist[hc] if args.diff_dist_per_head else args.dist          # :383-387
        # :388 verbatim -- torch.sort sorts the last (size-1) dim of the (K,1) tensor, i.e. this is
        # a scatter new[argsort[i]] = old[i] that also mutates args.dist in place.
        _K_dist[marginals_argsort] = torch.sort(_K_dist)[0]
    r = 1. / _K_dist                                                                    # :392
    r /= r.sum()                                                                        # :393
    L, logsum, info = sinkhorn(PS, r, args.lamb, N_global=Ng, group=group, backend=be)
    if group is not None:
        _allreduce(logsum, group)
    cost = -(1. / args.lamb) * float(logsum.item()) / Ng                                # :418-419
    if getattr(args, "rank", 0) == 0 and logger is not None:
        logger.info(f"error: {info['err']}, step : {info['iters']}")
        logger.info(f"opt took {(time.time() - tt) / 60.} min, {info['iters']} iters")
    optimize_L_sk_gpu.last_info = info
    return cost, L


optimize_L_sk = optimize_L_sk_gpu      # the name BASELINE.json's north_star uses
