"""Trace of sk_utils.cluster on 2 ranks vs 1 process (diagnostic): per head the inputs and outputs of the solve."""
import hashlib, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")


def h(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:10]


def worker(rank, world, port):
    import torch.distributed as dist
    from selavi_amd import sk_utils
    import tests.test_cluster_gpu as T
    orig = sk_utils.optimize_L_sk_gpu

    def traced(args, PS, hc, logger=None, group=None, N_global=None, backend=None):
        full = PS.clone()
        if group is not None:
            parts = [torch.empty_like(PS) for _ in range(world)]
            dist.all_gather(parts, PS)
            full = torch.cat(parts)
        hin = h(full)
        dist_in = None if args.dist is None else h(torch.cat([d.flatten() for d in args.dist]))
        cost, L = orig(args, PS, hc, logger, group=group, N_global=N_global, backend=backend)
        info = sk_utils.optimize_L_sk_gpu.last_info
        if rank == 0:
            print(f"[w{world}] head {hc}: PS {hin} N_local {PS.shape[0]} dist_in {dist_in} dist_out "
                  f"{h(torch.cat([d.flatten() for d in args.dist]))} iters {info['iters']} err {info['err']:.17g} cost {cost:.17g} "
                  f"alpha {h(info['alpha'])} L {h(L)}", flush=True)
        return cost, L
    traced.last_info = None
    sk_utils.optimize_L_sk_gpu = traced
    ret = {}
    T._cluster_worker(rank, world, port, 3, 8, ret, False)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 28444), nprocs=2, join=True)
    mp.spawn(worker, args=(1, 28666), nprocs=1, join=True)
