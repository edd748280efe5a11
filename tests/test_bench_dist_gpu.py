"""The driver's multi-GPU command, end to end, before the first 8-GPU node sees it.

`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2
--steps 3 --warmup 1` is what SCALE_rNN runs (with real RCCL, one rank per GPU).  The test boxes have ONE GPU and RCCL
refuses two ranks on a device, so the dry-run knobs of bench.py put both ranks on cuda:0 (SELAVI_BENCH_SHARE_GPU=1) with
gloo as the bootstrap and the library's own communicator paths forced on over the librccl test double
(SELAVI_NATIVE_COMM=force, SELAVI_RCCL_LIB): the WHOLE script -- data-parallel step with SyncBN + gradient buckets, the
preflight watchdog, forward timing, row-sharded SK through slv_sk_iterate_sharded, the cfg5 bf16 leg under data
parallelism, max-over-ranks timing, ONE JSON line from rank 0 -- runs as the driver will run it.
Reference wiring: /root/reference/main.py:117-118,156-160, utils.py:133-146."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run_oversubscribed(cmd, env, timeout):
    """Run the launcher; ONE retry if a rank died with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION.  Seen once in five runs of the
    8-rank rig (round 6: rank 7, six seconds after the rendezvous, the seven others healthy; three re-runs of the same tree
    green): eight processes time-sliced on ONE device is what this rig adds to the picture -- the driver saves and restores
    waves of kernels that fill the register file and LDS -- and not what a run with one process per GPU does.  Anything else
    fails at once, and a second death fails too."""
    for attempt in (0, 1):
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode == 0 or attempt == 1 or "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" not in p.stderr:
            return p
        sys.stderr.write("test_bench_dist_gpu: a rank of the oversubscribed rig died with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION; "
                         "retrying once\n")
    return p


def _check_comm(out, world, native):
    """The N > 1 line explains its exchanges: transport, ranks per communicator as RCCL counts them,
    SyncBN exchanges and their duration, bucket bytes and the exposed wait, the SK iteration split into pass and all-reduce."""
    c = out["comm"]
    assert c["world"] == world and c["preflight"] == "ok" and c["wrapper"] == "DataParallel"
    if native:
        assert c["transport"].startswith("native") and c["library"]
        tags = set(c["communicators"])
        assert {"bn", "bn_audio", "grad"} <= tags, tags
        for tag, d in c["communicators"].items():
            assert d["ranks_rccl_reports"] == world and d["world"] == world, (tag, d)
    else:
        assert c["transport"].startswith("torch.distributed") and c["communicators"] == {}
    sb, gr = c["syncbn"], c["grad_allreduce"]
    assert sb["exchanges_per_step"] >= 40 and sb["mean_ms"] > 0 and sb["total_ms_per_step"] > 0
    assert gr["collectives_per_step"] == 7 and gr["bytes_per_step"] > 170e6 and gr["exposed_wait_ms_per_step"] >= 0
    sk = c["sk"]
    assert sk["us_per_iter"] > 0 and sk["us_pass_reduce_update"] > 0 and sk["us_allreduce"] >= 0
    assert abs(sk["us_pass_reduce_update"] + sk["us_allreduce"] - sk["us_per_iter"]) <= 1e-6 * sk["us_per_iter"] or sk["us_allreduce"] == 0

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("transport", ["native_double", "torch_gloo"])
def test_driver_scale_command_two_ranks_on_one_gpu(transport):
    from tests.rccl_double.build import build_double
    env = dict(os.environ, SELAVI_BENCH_SHARE_GPU="1", SELAVI_BENCH_DIST_BACKEND="gloo", SLV_DBL_TIMEOUT_S="120",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if transport == "native_double":
        env.update(SELAVI_NATIVE_COMM="force", SELAVI_RCCL_LIB=build_double())
    else:
        env.pop("SELAVI_RCCL_LIB", None)
        env["SELAVI_NATIVE_COMM"] = "0"
    port = 24100 + os.getpid() % 500 + (0 if transport == "native_double" else 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--cfg5-batch", "16", "--cfg5-steps", "2", "--cfg5-warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2" and out["config"]["sync_bn"] is True
    assert out["value"] > 0 and abs(out["value"] - 32 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert out["sk"]["rows_per_gpu"] == 170752 // 2 and out["sk"]["iters_per_s"] > 0          # SK rows sharded over the ranks
    c5 = out["cfg5_bf16"]
    assert "error" not in c5 and c5["n_gpus"] == 2 and c5["config"]["parallelism"] == "dp2" and c5["value"] > 0
    assert out["cpu_baseline"] is None                                                         # rank 0 at N = 1 only
    assert "FAILED their preflight" not in p.stderr
    _check_comm(out, 2, native=(transport == "native_double"))


def test_driver_scale_command_eight_ranks_on_one_gpu():
    """The driver's `--gpus 8` command with all eight ranks on the one test GPU over the librccl double (per-GPU batch 2 so
    that eight copies of the step fit the box's patience): the host logic a real 8-GPU run executes -- 8-way SyncBN and
    buckets, the preflight over 8 ranks, Sinkhorn-Knopp rows in 8 shards of 21 344, the cfg5 leg under dp8, ONE JSON line."""
    from tests.rccl_double.build import build_double
    env = dict(os.environ, SELAVI_BENCH_SHARE_GPU="1", SELAVI_BENCH_DIST_BACKEND="gloo", SLV_DBL_TIMEOUT_S="300",
               HSA_ENABLE_IPC_MODE_LEGACY="0", SELAVI_NATIVE_COMM="force", SELAVI_RCCL_LIB=build_double(), SELAVI_BENCHMARK="0")
    port = 23100 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-native-leg", "--cfg5-batch", "2", "--cfg5-steps", "1", "--cfg5-warmup", "1"]
    p = _run_oversubscribed(cmd, env, 1100)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8" and out["config"]["sync_bn"] is True
    assert out["value"] > 0 and abs(out["value"] - 16 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert out["sk"]["rows_per_gpu"] == 21344 and out["sk"]["iters_per_s"] > 0
    c5 = out["cfg5_bf16"]
    assert "error" not in c5 and c5["n_gpus"] == 8 and c5["config"]["parallelism"] == "dp8" and c5["value"] > 0
    assert out["cpu_baseline"] is None
    assert "FAILED their preflight" not in p.stderr
    _check_comm(out, 8, native=True)
