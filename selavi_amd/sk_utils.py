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
import os
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

    def pass_reduce(self, P, N_global, beta, ws, grid):
        N, K = P.shape
        C.slv_sk_pass_reduce(ptr(P), N, N_global, K, ptr(beta), ptr(ws), grid, stream())

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


def _allreduce(t, group, comm=None):
    """Sum over the ranks of ``group``: through the RCCL communicator behind the C ABI when there is one (comm.py),
    else through torch.distributed."""
    if comm is not None and t.is_cuda and t.dtype in (torch.float64, torch.float32, torch.int64):
        comm.allreduce_(t)
        return
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _comm_of(group, backend):
    """The native communicator of ``group`` for the HIP backend (None: torch.distributed carries the exchange)."""
    if group is None or backend is not _HIP:
        return None
    from .comm import NativeComm
    return NativeComm.for_group(group)


def sinkhorn(P, r, lamb, N_global=None, group=None, backend=None, tol=1e-1, max_iter=2000,
             batch=10, grid=None, already_powered=False):
    """Run the SK loop on ``P`` (N_local x K fp64, destroyed).  Returns (labels, logsum, info).

    ``r``: K normalised marginals (sk_utils.py:392-393).  ``group``: torch.distributed process group
    whose ranks each hold a row shard (None = single GPU).  The loop follows sk_utils.py:400-406
    exactly: err is tested on counters 0,10,20,... only, so the iteration count is == 1 (mod 10)
    unless the 2000 cap hits."""
    be = backend or _HIP
    comm = _comm_of(group, be)
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
        _allreduce(be.s_view(ws, K, grid), group, comm)
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
            elif comm is not None:      # pass, local reduce, RCCL all-reduce, update x batch: one host call, one stream
                C.slv_sk_iterate_sharded(comm.h, ptr(P), N, N_global, K, ptr(beta), ptr(r), float(tol), int(max_iter),
                                         int(batch), ptr(ws), grid, stream())
            else:
                fused = getattr(be, "pass_reduce", None)        # one host call instead of two (optional in a backend)
                sv = be.s_view(ws, K, grid)
                for _ in range(batch):
                    if fused is not None:
                        fused(P, N_global, beta, ws, grid)
                    else:
                        be.pass_(P, N_global, beta, ws, grid)
                        be.local_reduce(K, ws, grid)
                    _allreduce(sv, group)                       # K col sums + err in one message
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
            _allreduce(colsum, group, _comm_of(group, be))
        marginals_argsort = torch.argsort(colsum)
        if (args.dist is None) or args.diff_dist_every:
            if args.distribution == 'gauss':
                if args.diff_dist_per_head:
                    _K_dists = [(torch.randn(size=(K, 1), dtype=torch.float64, device=dev)
                                 * args.gauss_sd + 1) * Ng / K for _ in range(args.headcount)]  # :371-373
                    if group is not None:
                        import torch.distributed as dist
                        for d in _K_dists:
                            dist.broadcast(d, src=_group_src(group), group=group)
                    args.dist = _K_dists
                    _K_dist = _K_dists[hc]
                else:
                    _K_dist = (torch.randn(size=(K, 1), dtype=torch.float64, device=dev)
                               * args.gauss_sd + 1) * Ng / K                          # :377
                    _K_dist = torch.clamp(_K_dist, min=1)                              # :378
                    if group is not None:
                        import torch.distributed as dist
                        dist.broadcast(_K_dist, src=_group_src(group), group=group)
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
        _allreduce(logsum, group, _comm_of(group, be))
    cost = -(1. / args.lamb) * float(logsum.item()) / Ng                                # :418-419
    if getattr(args, "rank", 0) == 0 and logger is not None:
        logger.info(f"error: {info['err']}, step : {info['iters']}")
        logger.info(f"opt took {(time.time() - tt) / 60.} min, {info['iters']} iters")
    optimize_L_sk_gpu.last_info = info
    return cost, L


optimize_L_sk = optimize_L_sk_gpu      # the name BASELINE.json's north_star uses


# ======================================================================================================
# match_order / get_cluster_assignments_gpu / cluster  (sk_utils.py:23-356,424-467)
# ======================================================================================================
def l1_cost_matrix(emb1, emb2, group=None, backend=None):
    """C[i][j] = sum_n |emb1[n, i] - emb2[n, j]| over ALL rows (all-reduced over the row shards)."""
    N, K = emb1.shape
    nsplit = max(1, min(64, N // 256))
    part = torch.empty(nsplit, K, K, dtype=torch.float64, device=emb1.device)
    out = torch.empty(K, K, dtype=torch.float64, device=emb1.device)
    C.slv_sk_l1_cost_matrix(ptr(emb1.contiguous()), ptr(emb2.contiguous()), N, K, ptr(part), nsplit, ptr(out),
                            stream())
    if group is not None:
        _allreduce(out, group, _comm_of(group, _HIP))
    return out


def _group_src(group):
    """Global rank of rank 0 of ``group`` (the ``src`` torch.distributed.broadcast wants)."""
    import torch.distributed as dist
    if group is None or group is dist.group.WORLD or not hasattr(dist, "get_global_rank"):
        return 0
    return dist.get_global_rank(group, 0)


def _hill_climb(Cm, steps, restarts, logger=None):
    """The random pair-swap search of sk_utils.py:436-461 on the K x K column-distance table (host).
    Consumes ``np.random.choice(K, 2, replace=False)`` exactly like the reference."""
    import numpy as np
    K = Cm.shape[0]
    cost0 = float(np.trace(Cm))
    best_cost, fin_perm = cost0, np.arange(K)
    if logger is not None:
        logger.info(f'initial cost: {cost0:.1f}')
    last_iter = 0                                   # (:441: not reset between restarts in the reference)
    for _ in range(restarts):
        perm = np.arange(K)                         # emb2 column currently sitting at position i
        for _iter in range(steps):
            i, j = np.random.choice(K, 2, replace=False)
            current = Cm[i, perm[i]] + Cm[j, perm[j]]
            future = Cm[i, perm[j]] + Cm[j, perm[i]]
            if current - future > 0:
                perm[i], perm[j] = perm[j], perm[i]
                last_iter = _iter
            if _iter - last_iter > 1000:
                break
        cost_try = float(Cm[np.arange(K), perm].sum())
        if logger is not None:
            logger.info(f"cost of this try: {cost_try:.2f}")
        if cost_try < best_cost:
            best_cost, fin_perm = cost_try, perm.copy()
    return fin_perm, best_cost


@torch.no_grad()
def match_order(args, emb1, emb2_in, W2, steps=50000, restarts=2, logger=None, group=None):
    """Drop-in for ``sk_utils.match_order`` (:424-467): align the audio head's clusters with the video
    head's by permuting the rows of the audio head's last Linear.  ``emb*`` may be row shards
    (``group``): the K x K L1 table is all-reduced, rank 0 searches, the permutation is broadcast."""
    import torch.distributed as dist
    K = emb1.shape[1]
    Cm = l1_cost_matrix(emb1, emb2_in, group=group)
    fin_perm = torch.arange(0, K, device=emb1.device)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    # the searching rank is rank 0 OF THE GROUP the shards live in (the reference searches on global rank 0, :431)
    searcher = dist.get_rank(group) == 0 if distributed else True
    if searcher:
        assert type(W2) == torch.nn.modules.linear.Linear or isinstance(W2, torch.nn.Linear)
        perm, best = _hill_climb(Cm.cpu().numpy(), steps, restarts, logger)
        fin_perm = torch.from_numpy(perm).to(emb1.device)
        if logger is not None:
            logger.info(f"final cost: {best:.2f}")
    if distributed:
        dist.broadcast(fin_perm, src=_group_src(group), group=group)
    W2.bias.data = W2.bias.data[fin_perm]
    W2.weight.data = W2.weight.data[fin_perm]
    return fin_perm


class _SubsetSequentialSampler(torch.utils.data.Sampler):
    """The indices of a rank's slice in order (the deterministic counterpart of SubsetRandomSampler)."""

    def __init__(self, indices):
        self.indices = indices

    def __iter__(self):
        return iter(int(i) for i in self.indices)

    def __len__(self):
        return len(self.indices)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


@torch.no_grad()
def get_cluster_assignments_gpu(args, dataset, model, logger=None, writer=None, group=None, iter_num=0):
    """Drop-in for ``sk_utils.get_cluster_assignments_gpu`` (:137-356), re-designed for sharded HBM:

    every rank runs the eval-mode feature pass over ITS contiguous dataset slice (:157-163), keeps the
    features it produced (no all_gather to rank 0, no per-batch barrier), applies each head to its shard
    with the MFMA GEMM, and the Sinkhorn-Knopp solve runs row-sharded over all ranks (one K+1 fp64
    all-reduce per iteration).  Labels are all-gathered at the end (replaces ``broadcast(L)``, :345-348).
    Returns L: N x headcount int64 on the device (rows beyond W*(N//W) stay zero, as in the reference)."""
    import numpy as np
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    grp = (group if group is not None else dist.group.WORLD) if distributed else None
    world = dist.get_world_size(grp) if distributed else 1
    rank = dist.get_rank(grp) if distributed else 0
    net = _unwrap(model)
    was_training = model.training
    model.eval()                                                                       # :150
    N = len(dataset)
    local_n = N // world                                                               # :158
    lo = rank * local_n
    dev = next(net.parameters()).device
    hc = args.headcount
    assert args.ind_groups <= hc                                                       # :183
    if hc > 1:
        net.return_features = True                                                     # :185-187
    L = torch.zeros((N, hc), dtype=torch.long, device=dev)
    order_heads = list(range(hc))
    np.random.shuffle(order_heads)                                                     # :191-192
    if distributed:
        # The reference shuffles on every rank too, but there only rank 0's order matters (rank 0 alone solves SK,
        # :287-329).  Here every rank solves its row shard of the SAME head, and the ranks' numpy streams diverge
        # (only the searching rank draws match_order's np.random.choice; dataset augmentation draws differ per
        # shard), so the order is rank 0's, broadcast.
        oh = torch.tensor(order_heads, dtype=torch.long, device=dev)
        dist.broadcast(oh, src=_group_src(grp), group=grp)
        order_heads = [int(h) for h in oh.tolist()]
    # :168 hard-codes 64.  Eval-mode outputs do not depend on the batch they are computed in, so a larger one only fills
    # the late layers better (bf16 pass: 5.8 k clips/s at 64, 6.5 k at 256); launch configurations -- and with them the
    # fp32 summation order -- follow the shape, so the default stays at the reference's value
    bs = int(getattr(args, "sk_batch_size", None) or os.environ.get("SELAVI_SK_BATCH", 64))
    idx_local = torch.arange(lo, lo + local_n)
    # :157-175: this rank's contiguous slice through a DataLoader (SubsetRandomSampler, args.workers decode workers,
    # pinned staging) so that a real decoding dataset does not serialise on the main process; shuffle_sk_pass=False
    # (tests) walks the slice in order instead
    shuffle = getattr(args, "shuffle_sk_pass", True)
    sampler = torch.utils.data.SubsetRandomSampler(idx_local) if shuffle else _SubsetSequentialSampler(idx_local)
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=bs, sampler=sampler,
                                             num_workers=int(getattr(args, "workers", 0) or 0),
                                             pin_memory=dev.type == "cuda", collate_fn=None)
    # opt-in: the feature pass in bf16 on the channels-last MFMA kernels (selavi_amd/infer16.py, ~3x faster; the
    # features are NOT the bit-exact fp32 ones -- the default stays fp32).  args.feature_pass or SELAVI_FEATURE_PASS
    # args.feature_pass / SELAVI_FEATURE_PASS -- "fp32" (default): the model's own eval forward, bit for bit what model.eval()
    # returns anywhere else; "fp32_folded": the fp32 trunks with BatchNorm folded into the weights for the length of the pass
    # (selavi_amd/infer32.py: conv + BN (+ shortcut) + ReLU in one launch, weight images made once per pass, the exact
    # three-piece operand split of the training path -- features 6e-7 off the plain forward, measured no faster: 2 131 against
    # 2 120 clips/s); "fp32x2" (opt-in): folded with two pieces per operand (half the matrix-core work: 2 800 clips/s; features
    # ~2e-4 off, labels not guaranteed identical); "bf16" (opt-in): selavi_amd/infer16.py, ~6 400 clips/s, features ~5e-3 off
    engine16 = None
    fp_mode = getattr(args, "feature_pass", None) or os.environ.get("SELAVI_FEATURE_PASS", "fp32")
    if fp_mode not in ("fp32", "fp32_folded", "fp32x2", "bf16"):
        raise ValueError(f"feature_pass {fp_mode!r}: fp32 | fp32_folded | fp32x2 | bf16")
    if fp_mode == "bf16":
        from . import infer16
        engine16 = infer16.Engine(net)
    import contextlib
    from . import infer32
    folded = (lambda: infer32.folded_eval(net, pieces=3 if fp_mode == "fp32_folded" else 2)) if fp_mode in ("fp32_folded", "fp32x2") \
        else contextlib.nullcontext
    for hd_grp_idx in range(args.ind_groups):                                          # :194
        # 1. feature pass over this rank's slice (every head group re-runs it: "decorrelated heads")
        bank_v = bank_a = None
        indices = torch.empty(local_n, dtype=torch.long, device=dev)
        fr = 0
        with folded():
          for batch in dataloader:                                                     # :196
            video, audio, idx = batch[0], batch[1], batch[3]                           # :198
            video = video.to(dev, non_blocking=True)                                   # :201-203
            audio = audio.to(dev, non_blocking=True)
            idx = idx.to(dev, non_blocking=True).long()
            if engine16 is None:
                feat_v, feat_a = model(video, audio)                                   # :206
            else:
                feat_v, feat_a = engine16.features(video, audio)
                if hc == 1:                                                            # :207-211: the bank holds logits
                    feat_v, feat_a = net.mlp_v.forward(feat_v), net.mlp_a.forward(feat_a)
                    if net.norm_feat:                                                  # model.py:236-239
                        feat_v = torch.nn.functional.normalize(feat_v, p=2, dim=1)
                        feat_a = torch.nn.functional.normalize(feat_a, p=2, dim=1)
            if feat_v.dim() == 1:
                feat_v, feat_a = feat_v.unsqueeze(0), feat_a.unsqueeze(0)
            if bank_v is None:
                bank_v = torch.empty(local_n, feat_v.shape[1], dtype=torch.float32, device=dev)
                bank_a = torch.empty_like(bank_v)
            to = fr + feat_v.shape[0]
            bank_v[fr:to], bank_a[fr:to], indices[fr:to] = feat_v, feat_a, idx
            fr = to
        heads = order_heads[hd_grp_idx::args.ind_groups]
        # 2. optional audio/video head alignment at the very first SK (:257-286)
        if args.match and iter_num == 0:
            for head in heads:
                if hc == 1:
                    head_a, lv, la = net.mlp_a, bank_v, bank_a          # hc == 1: banks already hold logits
                else:
                    head_a, head_v = getattr(net, f'mlp_a{head}'), getattr(net, f'mlp_v{head}')
                    lv, la = head_v.forward(bank_v), head_a.forward(bank_a)
                W2 = list(head_a.modules())[-1] if net.use_mlp else head_a
                match_order(args, softmax64(lv), softmax64(la), W2, steps=50000, restarts=2, logger=logger,
                            group=grp)
        # 3. Sinkhorn-Knopp per head, row-sharded (:300-327)
        costs = {}
        for head in heads:
            t0 = time.time()
            if hc == 1:
                lv, la = bank_v, bank_a
            else:
                lv = getattr(net, f'mlp_v{head}').forward(bank_v)                       # :309-312
                la = getattr(net, f'mlp_a{head}').forward(bank_a)
            PS = head_probabilities(lv, la)                                            # :309-315 fused
            cost, L_head = optimize_L_sk_gpu(args, PS, head, logger, group=grp, N_global=local_n * world)
            L[indices, head] = L_head                                                   # :323 (local rows)
            costs[head] = cost
            if logger is not None and rank == 0:
                logger.info(f"Head {head}, Cost: (video): {cost:.3f}; time: {time.time() - t0:.3f}")
        if logger is not None and rank == 0 and costs:
            logger.info(f"Final Cost: (video): {np.mean(list(costs.values())):.3f}")
        if writer and costs:
            writer.add_scalar('train/LP-cost', np.mean(list(costs.values())), iter_num)
    if distributed:
        # each rank filled only its own rows (disjoint): a sum all-reduce assembles L everywhere
        dist.all_reduce(L, op=dist.ReduceOp.SUM, group=grp)
    net.return_features = False                                                        # :354
    model.train(was_training or True)                                                  # :355
    return L


def cluster(args, selflabels, dataset, model, sk_counter, logger, writer, group, iter_num):
    """Drop-in for ``sk_utils.cluster`` (:23-134): one SK round + NMI logging (sklearn on the host, as
    in the reference).  Returns the new ``selflabels`` (N x headcount int64, device)."""
    import numpy as np
    from sklearn.metrics.cluster import adjusted_mutual_info_score, normalized_mutual_info_score
    selflabels_old = selflabels.clone()
    with torch.no_grad():
        selflabels = get_cluster_assignments_gpu(args, dataset, model, logger, writer, group, iter_num)
    self_labels_np = selflabels[:, 0].cpu().numpy()
    sk_counter += 1                                                                     # :42 (local, as in the reference)
    rank0 = getattr(args, "rank", 0) == 0
    nmi_v = normalized_mutual_info_score(self_labels_np, selflabels_old[:, 0].cpu().numpy(),
                                         average_method='arithmetic')
    if rank0 and logger is not None:
        logger.info(f'NMI_v: {nmi_v}')
    if writer:
        writer.add_scalar('train/nmi_v/iter', nmi_v, iter_num)
        writer.add_scalar('train/optim_count/iter', sk_counter, iter_num)
    if hasattr(dataset, "_labels"):
        true_labels = np.array(dataset._labels)[dataset.valid_indices]
        nmi_l = normalized_mutual_info_score(self_labels_np, true_labels, average_method='arithmetic')
        anmi_l = adjusted_mutual_info_score(self_labels_np, true_labels, average_method='arithmetic')
        if rank0 and logger is not None:
            logger.info(f"NMI-tolabels: {nmi_l}")
            logger.info(f"aNMI-tolabels: {anmi_l}")
        if writer:
            writer.add_scalar('train/nmi-tolabels_v/iter', nmi_l, iter_num)
            writer.add_scalar('train/a-nmi-tolabels_v/iter', anmi_l, iter_num)
        if sk_counter % 10 == 0:                                                        # :89-122
            from scipy.stats import entropy
            ents, purs = [], []
            for lab in np.unique(self_labels_np):
                sel = self_labels_np == lab
                if sel.sum() != 0:
                    _, counts = np.unique(true_labels[sel], return_counts=True)
                    purs.append(max(counts) / sum(1.0 * counts))
                    ents.append(entropy(counts / sum(1.0 * counts)))
            if logger is not None:
                logger.info(f"Avg entropy: {np.mean(ents)}")
                logger.info(f"Avg purity: {np.mean(purs)}")
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier(group=group) if group is not None else dist.barrier()
    cluster.last_nmi = nmi_v
    return selflabels
