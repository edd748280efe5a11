"""Which two-rank rig is not reproducible run to run: torch.distributed/gloo exchanges, or the library's own communicators
over the librccl test double?  Spawns the same 3-step two-rank job several times per rig and compares state hashes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.multiprocessing as mp

def main():
    from tests.test_cluster_gpu import _ddp_worker
    from tests.test_native_comm_gpu import build_double
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    for rig in ("gloo", "native"):
        if rig == "native":
            os.environ["SELAVI_RCCL_LIB"] = build_double()
            os.environ["SELAVI_NATIVE_COMM"] = "force"
            os.environ["SLV_DBL_TIMEOUT_S"] = "90"
        runs = []
        for i in range(n):
            ret = mp.Manager().dict()
            mp.spawn(_ddp_worker, args=(2, 24000 + (os.getpid() + 37 * i + (500 if rig == "native" else 0)) % 900, ret, "native", prec), nprocs=2, join=True)
            runs.append(dict(ret[0][2]))
        bad = [sorted(k for k in runs[0] if runs[i][k] != runs[0][k]) for i in range(1, n)]
        print(f"{rig} rig, {prec}: runs differing from run 0: {[len(b) for b in bad]}; first names: {[b[:2] for b in bad if b][:2]}", flush=True)

if __name__ == "__main__":
    main()
