"""CPU restatement of the Sinkhorn-Knopp pseudo-label solver (TEST INFRASTRUCTURE ONLY).

Follows ``/root/reference/src/sk_utils.py``:
  * ``optimize_L_sk``        <- ``optimize_L_sk_gpu``            (sk_utils.py:359-422)
  * ``head_probabilities``   <- per-head softmax64 product        (sk_utils.py:300-315)
  * ``match_order_ref``      <- ``match_order`` with the swap sequence injected (sk_utils.py:424-467)
  * ``sk_schedule``          <- main.py:163-171

numpy fp64 throughout (the reference runs torch fp64; IEEE double either way).  Parity is pinned
by ``tests/golden/sk_*.npz`` which were produced by executing the reference's own
``optimize_L_sk_gpu`` in the build container (``tests/golden/make_golden.py``).
"""
import numpy as np


def softmax64(x):
    """``torch.nn.functional.softmax(x, dim=1, dtype=torch.float64)`` (sk_utils.py:208-211)."""
    x = np.asarray(x).astype(np.float64)
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=1, keepdims=True)


def head_probabilities(logits_v, logits_a):
    """PS = softmax64(head_v(feat_v)) * softmax64(head_a(feat_a))   (sk_utils.py:309-315)."""
    return softmax64(logits_v) * softmax64(logits_a)


def marginals(K, N, PS=None, distribution='default', dist=None):
    """Cluster-size vector ``_K_dist`` (K,) as sk_utils.py:366-388 leaves it.

    ``dist`` is the already-sampled Gaussian size vector for this head (the reference draws it
    with torch.randn on the GPU, sk_utils.py:372/377 -- not portable, so fixtures pass it in).
    NOTE sk_utils.py:388 reads ``_K_dist[marginals_argsort] = torch.sort(_K_dist)[0]`` but ``_K_dist``
    is (K, 1) and ``torch.sort`` sorts the LAST dim (size 1) -> the "sort" is a no-op and the line is
    a plain scatter ``new[argsort[i]] = old[i]`` that also mutates ``args.dist`` in place.  Restated
    literally (verified against the executed reference, tests/golden/sk_gauss_per_head.npz);
    the returned array is that mutated vector.
    """
    if distribution == 'default':
        return np.ones(K, dtype=np.float64)
    assert dist is not None and PS is not None
    kd = np.array(dist, dtype=np.float64).reshape(K).copy()
    order = np.argsort(PS.sum(0), kind='stable')
    old = kd.copy()
    kd[order] = old                      # literal :388 (sort over a size-1 dim == identity)
    return kd


def optimize_L_sk(PS, lamb=20, K_dist=None, max_iter=2000, tol=1e-1, check_every=10):
    """Restatement of ``optimize_L_sk_gpu`` (sk_utils.py:359-422) for a given marginal vector.

    PS: (N, K) float64 joint probabilities (NOT yet raised to lamb/2).  Returns
    ``(cost, newL, info)`` with info = dict(iters, err, alpha, beta).
    """
    PS = np.array(PS, dtype=np.float64)          # private copy: the reference destroys its input
    N, K = PS.shape
    kd = np.ones(K) if K_dist is None else np.asarray(K_dist, dtype=np.float64).reshape(K)
    beta = np.ones((N, 1)) / N                   # :390
    np.power(PS, 0.5 * lamb, out=PS)             # :391
    r = (1.0 / kd).reshape(K, 1)                 # :392
    r /= r.sum()                                 # :393
    c = 1.0 / N
    err, counter = 1e6, 0
    alpha = None
    import time
    t_loop = time.perf_counter()                 # (bench.py's cpu_baseline of the SK half times this loop)
    while err > tol and counter < max_iter:      # :400
        alpha = r / (beta.T @ PS).T              # :401
        beta_new = c / (PS @ alpha)              # :402
        if counter % check_every == 0:           # :403
            err = float(np.sum(np.abs(beta.squeeze() / beta_new.squeeze() - 1.0)))
        beta = beta_new
        counter += 1
    t_loop = time.perf_counter() - t_loop
    PS *= beta                                   # :411
    PS *= alpha.T                                # :412
    newL = np.argmax(PS, 1)                      # :413
    PS *= (1.0 / alpha).T                        # :416
    PS *= 1.0 / beta                             # :417
    with np.errstate(divide='ignore', invalid='ignore'):
        sol = np.nansum(np.log(PS[np.arange(N), newL]))   # :418
    cost = -(1.0 / lamb) * sol / N               # :419
    return cost, newL.astype(np.int64), dict(iters=counter, err=err, alpha=alpha.ravel().copy(),
                                             beta=beta.ravel().copy(), loop_s=t_loop)


def optimize_L_sk_sharded(PS, world, lamb=20, K_dist=None, max_iter=2000, tol=1e-1):
    """W-rank row-sharded emulation: rows split contiguously, the K-vector
    ``beta^T PS`` is summed over ranks in rank order each iteration, ``err`` on tested iterations."""
    PS = np.array(PS, dtype=np.float64)
    N, K = PS.shape
    kd = np.ones(K) if K_dist is None else np.asarray(K_dist, dtype=np.float64).reshape(K)
    np.power(PS, 0.5 * lamb, out=PS)
    bounds = [(r * N) // world for r in range(world + 1)]
    shards = [PS[bounds[r]:bounds[r + 1]] for r in range(world)]
    betas = [np.ones((s.shape[0], 1)) / N for s in shards]
    r = (1.0 / kd).reshape(K, 1)
    r /= r.sum()
    c = 1.0 / N
    err, counter = 1e6, 0
    while err > tol and counter < max_iter:
        s = np.zeros((1, K))
        for sh, b in zip(shards, betas):
            s = s + b.T @ sh
        alpha = r / s.T
        new = [c / (sh @ alpha) for sh in shards]
        if counter % 10 == 0:
            err = float(sum(np.sum(np.abs(b.squeeze(1) / n.squeeze(1) - 1.0)) for b, n in zip(betas, new)))
        betas = new
        counter += 1
    newL = np.concatenate([np.argmax((sh * b) * alpha.T, 1) for sh, b in zip(shards, betas)])
    return newL.astype(np.int64), dict(iters=counter, err=err, alpha=alpha.ravel().copy())


def match_order_ref(emb1, emb2_in, swaps, patience=1000):
    """``match_order`` hill-climb (sk_utils.py:436-461) for ONE restart with the candidate pair
    sequence ``swaps`` (list of (i, j)) given explicitly instead of ``np.random.choice``.
    Returns (perm, cost)."""
    emb1 = np.asarray(emb1, dtype=np.float64)
    emb2 = np.array(emb2_in, dtype=np.float64)
    K = emb1.shape[1]
    perm = np.arange(K)
    last = 0

    def c(a, b):
        return np.abs(a - b).sum()
    for it, (i, j) in enumerate(swaps):
        current = c(emb1[:, i], emb2[:, i]) + c(emb1[:, j], emb2[:, j])
        future = c(emb1[:, i], emb2[:, j]) + c(emb1[:, j], emb2[:, i])
        if current - future > 0:
            emb2[:, [i, j]] = emb2[:, [j, i]]
            perm[i], perm[j] = perm[j], perm[i]
            last = it
        if it - last > patience:
            break
    return perm, c(emb1, np.asarray(emb2_in)[:, perm])


def sk_schedule(epochs, n_batches, nopts=100, schedulepower=1.5):
    """main.py:168-170: iteration indices at which an SK round fires (popped from the end)."""
    sched = (epochs * n_batches * (np.linspace(0, 1, nopts) ** schedulepower)[::-1]).tolist()
    return [(epochs + 2) * n_batches] + sched
