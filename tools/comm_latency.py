"""Latency of ONE SyncBN exchange on the compute stream (world of one rank over RCCL: every launch, no wire time):
  single   slv_bn_stats_finalize                                  (no exchange: the single-process kernel)
  native   slv_bn_sync_finalize = partials -> sums, ncclAllReduce, finalize in one library call (selavi_amd/comm.py)
  torch    the same three steps with torch.distributed.all_reduce in the middle (process-group stream + event hops)
and of the raw all-reduce of 2C doubles.  Usage: python tools/comm_latency.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29778")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from selavi_amd import ops
from selavi_amd.comm import NativeComm
comm = NativeComm.for_group(None)
dev = torch.device("cuda:0")
Cc, nblk = 144, 6272
g = torch.Generator(device=dev).manual_seed(1)
ps, pq = torch.randn(Cc, nblk, device=dev, generator=g), torch.rand(Cc, nblk, device=dev, generator=g)
gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


sums = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
rows = [("single  (slv_bn_stats_finalize)", lambda: ops.bn_train_finalize(ps, pq, 1e5, gamma, beta, rm, rv, 0.1, 1e-5)),
        ("native  (slv_bn_sync_finalize)", lambda: ops.bn_train_finalize(ps, pq, 1e5, gamma, beta, rm, rv, 0.1, 1e-5, sync=(comm, 1))),
        ("torch   (kernel, dist.all_reduce, kernel)", lambda: ops.bn_train_finalize(ps, pq, 1e5, gamma, beta, rm, rv, 0.1, 1e-5, sync=(None, 1))),
        ("raw all-reduce 2C fp64, native", lambda: comm.allreduce_(sums)),
        ("raw all-reduce 2C fp64, torch.distributed", lambda: dist.all_reduce(sums))]
print(f"librccl: {comm.library()}   (us per call, host-enqueue bound where the GPU work is shorter; {reps} reps)")
for name, fn in rows:
    print(f"{name:46s} {timeit(fn):8.1f} us")
dist.destroy_process_group()
