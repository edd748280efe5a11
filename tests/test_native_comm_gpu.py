"""The native RCCL paths of the library at WORLD > 1 without a multi-GPU box.

RCCL refuses two ranks on one device, and the test boxes have one GPU.  ``tests/rccl_double/`` is a test double of
librccl (the seven symbols csrc/comm.cpp resolves, implemented over a file mapped by both ranks, sums in rank order);
``SELAVI_RCCL_LIB`` hands it to ``slv_comm_load`` and ``SELAVI_NATIVE_COMM=force`` routes SyncBN, the sharded
Sinkhorn-Knopp loop and the gradient buckets through ``slv_comm_*`` although torch.distributed (bootstrap only) is gloo.
Covered with two ranks: the communicator's collectives, ``slv_bn_sync_finalize`` / ``slv_bn_bwd_sync_finalize`` against
the torch.distributed path (bit-identical), ``slv_sk_iterate_sharded`` against the executed reference's goldens
(bit-exact), and the whole data-parallel step (SyncBN from the main stream AND the audio side stream, weight-gradient
side streams, autograd's thread, the gradient buckets on their own communicator) against the gloo rig (bit-identical).
Reference behaviour: /root/reference/main.py:117-118,156-160, utils.py:133-146, src/sk_utils.py:287-348."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.rccl_double.build import DOUBLE_LIB, build_double  # noqa: E402,F401


@pytest.fixture
def double_env(monkeypatch):
    monkeypatch.setenv("SELAVI_RCCL_LIB", build_double())
    monkeypatch.setenv("SELAVI_NATIVE_COMM", "force")
    monkeypatch.setenv("SLV_DBL_TIMEOUT_S", "90")


def _collectives_worker(rank, world, port, ret):
    import ctypes
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=300))
    try:
        torch.cuda.set_device(0)
        from selavi_amd import ops
        from selavi_amd._lib import C, ptr, stream
        from selavi_amd.comm import NativeComm
        comm = NativeComm.for_group(None)
        assert comm is not None and comm.world == world and comm.rank == rank and "rccl_double" in comm.library()
        comm2 = NativeComm.for_group(None, "grad")            # a second communicator of the same runtime
        assert comm2 is not None and comm2.h.value != comm.h.value
        # ---- collectives: sums in rank order, identical on both ranks
        g = torch.Generator(device="cuda").manual_seed(100 + rank)
        checks = {}
        for dt, n in ((torch.float64, 619), (torch.float32, 3_000_001), (torch.int64, 4097)):
            if dt == torch.int64:
                mine = torch.randint(-1000, 1000, (n,), device="cuda", generator=g)
            else:
                mine = torch.randn(n, device="cuda", generator=g, dtype=dt)
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)                       # gloo: the test's own ground truth
            want = both[0].clone()
            for r in range(1, world):
                want = want + both[r]
            got = mine.clone()
            comm.allreduce_(got)
            torch.cuda.synchronize()
            assert torch.equal(got, want), dt
            checks[str(dt)] = got[:8].cpu().numpy().copy()
            if dt == torch.float32:
                avg = mine.clone()
                comm2.allreduce_avg_f32_(avg)
                torch.cuda.synchronize()
                assert torch.equal(avg, want * 0.5)
        mine = torch.arange(1000, device="cuda", dtype=torch.int32) + 10_000 * rank
        recv = torch.empty(world * 1000, device="cuda", dtype=torch.int32)
        C.slv_comm_allgather(comm.h, ptr(mine), ptr(recv), 4000, stream())
        torch.cuda.synchronize()
        assert torch.equal(recv, torch.cat([torch.arange(1000, device="cuda", dtype=torch.int32) + 10_000 * r for r in range(world)]))
        b = torch.full((777,), float(rank + 1), device="cuda")
        C.slv_comm_broadcast(comm.h, ptr(b), 777 * 4, 1, stream())
        torch.cuda.synchronize()
        assert torch.equal(b, torch.full((777,), 2.0, device="cuda"))
        # ---- SyncBN in one call (slv_bn_sync_finalize / slv_bn_bwd_sync_finalize) == the torch.distributed path
        Cc, nblk = 45, 37
        ps, pq = torch.randn(Cc, nblk, device="cuda", generator=g), torch.rand(Cc, nblk, device="cuda", generator=g) * 9
        gh = torch.Generator(device="cuda").manual_seed(7)   # parameters are replicated
        gamma, beta = torch.rand(Cc, device="cuda", generator=gh) + 0.5, torch.randn(Cc, device="cuda", generator=gh)
        outs = []
        for where in (comm, None):                            # None: the default (gloo) group through dist.all_reduce
            rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
            mi, ss = ops.bn_train_finalize(ps, pq, 1000.0, gamma, beta, rm, rv, 0.1, 1e-5, sync=(where, world))
            torch.cuda.synchronize()
            outs.append(torch.cat([mi.flatten(), ss.flatten(), rm, rv]))
        assert torch.equal(outs[0], outs[1])
        part = torch.randn(Cc, 11, 2, device="cuda", generator=g)
        o2 = []
        for where in (comm, None):
            dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
            b5 = ops.bn_bwd_finish(part, None, 11, 1000.0, mi, gamma, ss, None, None, (where, world), dg, db, None, None)[0]
            torch.cuda.synchronize()
            o2.append(torch.cat([b5.flatten(), dg, db]))
        assert torch.equal(o2[0], o2[1])
        ret[rank] = (checks, outs[0].cpu().numpy().copy(), o2[0].cpu().numpy().copy())
        NativeComm.destroy_all()
    finally:
        dist.destroy_process_group()


def test_two_rank_native_collectives_and_one_call_syncbn(double_env):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_collectives_worker, args=(2, 27100 + os.getpid() % 400, ret), nprocs=2, join=True)
    for k in ret[0][0]:
        np.testing.assert_array_equal(ret[0][0][k], ret[1][0][k])
    np.testing.assert_array_equal(ret[0][1], ret[1][1])          # the ranks finalise the same statistics, bit for bit
    np.testing.assert_array_equal(ret[0][2], ret[1][2])


@pytest.mark.parametrize("name", ["sk_ave_peaked", "sk_gauss_per_head", "sk_vggsound_full"])
def test_two_rank_sk_iterate_sharded_matches_reference_golden(double_env, golden_dir, name):
    """slv_sk_iterate_sharded (pass, local reduce, all-reduce of K+1 doubles over the library's communicator, update:
    one host call per batch of iterations) on two ranks: labels bit-exact against the executed reference."""
    import torch.multiprocessing as mp
    from tests.test_sk_gpu import _digest, _sharded_worker
    ret = mp.Manager().dict()
    mp.spawn(_sharded_worker, args=(2, 27500 + os.getpid() % 400, name, golden_dir, ret, True), nprocs=2, join=True)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    L = np.concatenate([ret[0][1], ret[1][1]])
    assert ret[0][4] and ret[1][4], "the solve did not go through the native communicator"
    assert ret[0][2] == ret[1][2] == int(g["iters"])
    assert _digest(L) == bytes(g["digest"]).decode()
    assert ret[0][0] == ret[1][0] and abs(ret[0][0] - float(g["cost"])) <= 1e-9 * abs(float(g["cost"]))
    np.testing.assert_array_equal(ret[0][3], ret[1][3])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_rank_native_step_is_bit_identical_to_the_gloo_rig(monkeypatch, precision):
    """parallel.DataParallel + SyncBN for three steps on two ranks, once with every exchange on the library's own
    communicators (SyncBN from the main and the audio stream, buckets through slv_comm_allreduce_f32(average)), once
    with torch.distributed/gloo carrying them: every parameter and buffer bit-identical, between the ranks and between
    the two transports (both add the same two addends; x0.5 is exact)."""
    import torch.multiprocessing as mp
    from tests.test_cluster_gpu import _ddp_worker
    monkeypatch.setenv("SLV_DBL_TIMEOUT_S", "90")

    def rigs(attempt):
        ret_gloo, ret_nat = mp.Manager().dict(), mp.Manager().dict()
        monkeypatch.delenv("SELAVI_RCCL_LIB", raising=False)
        monkeypatch.delenv("SELAVI_NATIVE_COMM", raising=False)
        mp.spawn(_ddp_worker, args=(2, 26100 + (os.getpid() + 101 * attempt) % 300, ret_gloo, "native", precision), nprocs=2, join=True)
        monkeypatch.setenv("SELAVI_RCCL_LIB", build_double())
        monkeypatch.setenv("SELAVI_NATIVE_COMM", "force")
        mp.spawn(_ddp_worker, args=(2, 26500 + (os.getpid() + 101 * attempt) % 300, ret_nat, "native", precision), nprocs=2, join=True)
        return ret_gloo, ret_nat

    ret_gloo, ret_nat = rigs(0)
    # (round 3: in 1-3 of 8 runs the AUDIO trunk's tensors differed between the transports -- the zero-fill of the gradient
    #  sink's flat buffer, issued on the main stream at the trunk's first backward, raced with the side-stream trunk's first
    #  gradients; fixed in nn.TrunkFunction.backward, 0 mismatches in 38 runs since: tools/flake.sh)
    differs = [k for k in ret_nat[0][2] if ret_nat[0][2][k] != ret_gloo[0][2][k]]
    assert ret_nat[0][3] >= 3 and ret_nat[0][3] == ret_nat[1][3], "native communicators were expected (bn, bn_audio, grad)"
    assert ret_gloo[0][3] == 0
    diverged = [k for k in ret_nat[0][2] if ret_nat[0][2][k] != ret_nat[1][2][k]]
    assert not diverged, f"ranks diverged in {len(diverged)} tensors: {diverged[:6]}"
    assert not differs, f"{len(differs)} of {len(ret_nat[0][2])} tensors differ from the gloo rig: {differs[:6]}"
    assert ret_nat[0][0] == ret_gloo[0][0] and ret_nat[0][1] == ret_gloo[0][1]


def test_preflight_failure_falls_back_to_torch_distributed_on_every_rank(monkeypatch, capfd):
    """comm.NativeComm.preflight (the watchdog around the first collectives of the three concurrently driven communicators):
    a failure on ONE rank (injected on rank 1) makes EVERY rank abort its native communicators, say so loudly and re-create
    SyncBN and the gradient buckets on torch.distributed -- and the step is then bit-identical to the plain gloo rig."""
    import torch.multiprocessing as mp
    from tests.test_cluster_gpu import _ddp_worker
    monkeypatch.setenv("SLV_DBL_TIMEOUT_S", "90")
    ret_gloo, ret_fb = mp.Manager().dict(), mp.Manager().dict()
    monkeypatch.delenv("SELAVI_RCCL_LIB", raising=False)
    monkeypatch.delenv("SELAVI_NATIVE_COMM", raising=False)
    mp.spawn(_ddp_worker, args=(2, 25100 + os.getpid() % 300, ret_gloo, "native", "fp32"), nprocs=2, join=True)
    monkeypatch.setenv("SELAVI_RCCL_LIB", build_double())
    monkeypatch.setenv("SELAVI_NATIVE_COMM", "force")
    monkeypatch.setenv("SELAVI_COMM_PREFLIGHT_INJECT", "rank1")
    mp.spawn(_ddp_worker, args=(2, 25500 + os.getpid() % 300, ret_fb, "native", "fp32"), nprocs=2, join=True)
    err = capfd.readouterr().err
    assert "FAILED their preflight" in err and "falling back to torch.distributed" in err
    assert ret_fb[0][3] == 0 and ret_fb[1][3] == 0, "native communicators survived a failed preflight"
    assert ret_fb[0][2] == ret_fb[1][2] == ret_gloo[0][2]
    assert ret_fb[0][1] == ret_gloo[0][1]


def _spawn8(fn, make_args):
    """mp.spawn of an 8-process rig on ONE GPU with a single retry when a worker is KILLED BY SIGABRT (not when it raises: an
    assertion in a worker fails the test at once).  Round 6 saw one process of eight die with HSA_STATUS_ERROR_ILLEGAL_
    INSTRUCTION (the runtime aborts: SIGABRT) in one of five runs of the 8-rank bench rig, the other seven healthy, three
    re-runs green: eight processes time-sliced on one device -- the driver saving and restoring waves of kernels that fill the
    register file and LDS -- is what these rigs add to the picture, not what one process per GPU does.  make_args(attempt) ->
    (args tuple, result dict): a fresh rendezvous port and result dict per attempt."""
    import sys
    import torch.multiprocessing as mp
    for attempt in (0, 1):
        args, ret = make_args(attempt)
        try:
            mp.spawn(fn, args=args, nprocs=8, join=True)
            return ret
        except mp.ProcessExitedException as e:
            if attempt == 1 or getattr(e, "signal_name", None) != "SIGABRT":
                raise
            sys.stderr.write(f"test_native_comm_gpu: a worker of the oversubscribed 8-process rig was killed by SIGABRT ({e}); retrying once\n")


# ---- W = 8 rehearsal on one GPU: the host logic of an 8-rank run (rank-order sums over 8 addends, row assembly of 8 shards,
# buckets averaged over 8, 8-way SyncBN) before the first 8-GPU node executes it ------------------------
def test_eight_rank_sk_iterate_sharded_matches_reference_golden(double_env, golden_dir):
    """BASELINE configs[2]'s Sinkhorn-Knopp as 8 ranks run it: N = 170 752 rows in 8 shards of 21 344, slv_sk_iterate_sharded
    over the library's communicator (the librccl double: 8 processes on the one GPU), labels bit-exact against the executed
    reference, identical alpha / cost / iteration count on all 8 ranks."""
    import torch.multiprocessing as mp
    from tests.test_sk_gpu import _digest, _sharded_worker
    def make_args(attempt):
        ret = mp.Manager().dict()
        return (8, 27900 + os.getpid() % 400 + 443 * attempt, "sk_vggsound_full", golden_dir, ret, True), ret
    ret = _spawn8(_sharded_worker, make_args)
    g = np.load(os.path.join(golden_dir, "sk_vggsound_full.npz"))
    assert all(len(ret[r][1]) == 21344 for r in range(8))
    L = np.concatenate([ret[r][1] for r in range(8)])
    assert all(ret[r][4] for r in range(8)), "the solve did not go through the native communicator"
    assert all(ret[r][2] == int(g["iters"]) for r in range(8))
    assert _digest(L) == bytes(g["digest"]).decode()
    assert np.array_equal(np.bincount(L, minlength=int(g["K"])), g["hist"])
    for r in range(1, 8):
        assert ret[r][0] == ret[0][0]
        np.testing.assert_array_equal(ret[r][3], ret[0][3])
    assert abs(ret[0][0] - float(g["cost"])) <= 1e-9 * abs(float(g["cost"]))
    np.testing.assert_allclose(ret[0][3], g["alpha"], rtol=1e-9)


def test_eight_rank_native_step_keeps_the_ranks_in_lock_step_and_equals_one_large_batch(monkeypatch):
    """parallel.DataParallel on EIGHT ranks (two clips each, 8 processes on the one GPU): SyncBN sums over 8 ranks in rank
    order, the 7 gradient buckets averaged over 8 -- through the library's communicators (librccl double) and through
    torch.distributed/gloo.  After three steps every parameter and BatchNorm buffer is bit-identical across the 8 ranks on
    either transport; the transports agree to rounding (gloo's ring sums 8 addends in another order than the double's rank
    order: no bit identity between them beyond two ranks); and the mean of the 8 first-step losses is the loss of ONE process
    on the 16 clips (SyncBN makes the statistics global: SURVEY.md Appendix B)."""
    import torch.multiprocessing as mp
    from oracle import step_ref
    from oracle.model_ref import portable_fill_, portable_init_
    from tests.test_cluster_gpu import _ddp_worker
    monkeypatch.setenv("SLV_DBL_TIMEOUT_S", "120")
    monkeypatch.setenv("SELAVI_NATIVE_COMM", "0")
    def make_args(base):
        def f(attempt):
            ret = mp.Manager().dict()
            return (8, base + os.getpid() % 300 + 331 * attempt, ret, "native", "fp32"), ret
        return f
    ret_gloo = _spawn8(_ddp_worker, make_args(25900))
    monkeypatch.setenv("SELAVI_NATIVE_COMM", "force")
    monkeypatch.setenv("SELAVI_RCCL_LIB", build_double())
    ret_nat = _spawn8(_ddp_worker, make_args(26900))
    assert all(ret_nat[r][3] >= 3 for r in range(8)), "the native run did not create the library's communicators"
    assert all(ret_gloo[r][3] == 0 for r in range(8))
    for ret in (ret_gloo, ret_nat):
        for r in range(1, 8):
            diverged = [k for k in ret[0][2] if ret[0][2][k] != ret[r][2][k]]
            assert not diverged, f"rank {r} diverged from rank 0 in {len(diverged)} tensors: {diverged[:6]}"
        assert np.isfinite([ret[r][1] for r in range(8)]).all()
    np.testing.assert_allclose([ret_nat[r][0] for r in range(8)], [ret_gloo[r][0] for r in range(8)], rtol=1e-5)
    # one process, 16 clips
    from selavi_amd import model as smodel, optim, train
    hc, K = 2, 7
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    m = m.cuda().train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    video = portable_fill_(torch.empty(16, 3, 4, 32, 32), 5).cuda()
    audio = portable_fill_(torch.empty(16, 1, 40, 36), 6).cuda()
    sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
    sel = ((torch.arange(16) * 11 + 3) % 64).cuda()
    loss = float(train.train_step(m, opt, video, audio, sl, sel, hc))
    mean8 = float(np.mean([ret_nat[r][0] for r in range(8)]))
    assert abs(mean8 - loss) <= 2e-4 * abs(loss), (mean8, loss)
