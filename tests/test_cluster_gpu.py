"""GPU tests of the SK round orchestration (sk_utils.cluster / get_cluster_assignments_gpu / match_order)
and of the data-parallel wiring (DDP gradient averaging + SyncBN) on one GPU with two gloo ranks."""
import os

import numpy as np
import pytest
import torch

from oracle import sk_ref, step_ref
from oracle.model_ref import portable_fill_, portable_init_

pytestmark = pytest.mark.gpu


class Args:
    def __init__(self, **kw):
        self.distribution, self.dist, self.diff_dist_every = 'default', None, False
        self.diff_dist_per_head, self.gauss_sd, self.headcount = True, 0.1, 1
        self.lamb, self.rank, self.ind_groups, self.match = 20, 0, 1, False
        self.shuffle_sk_pass = False
        self.__dict__.update(kw)


def test_l1_cost_matrix_and_match_order_golden(golden_dir):
    from selavi_amd import sk_utils
    g = np.load(os.path.join(golden_dir, "match_order.npz"))
    e1, e2 = torch.from_numpy(g["emb1"]).cuda(), torch.from_numpy(g["emb2"]).cuda()
    Cm = sk_utils.l1_cost_matrix(e1, e2).cpu().numpy()
    want = np.abs(g["emb1"][:, :, None] - g["emb2"][:, None, :]).sum(0)
    np.testing.assert_allclose(Cm, want, rtol=1e-12)
    K = e1.shape[1]
    W2 = torch.nn.Linear(16, K).cuda()
    W2.weight.data = torch.from_numpy(g["w_before"]).cuda()
    W2.bias.data = torch.from_numpy(g["b_before"]).cuda()
    it = iter(list(g["pairs"]))
    orig = np.random.choice
    np.random.choice = lambda *a, **k: next(it)
    try:
        # the fixture was recorded with steps=3000
        sk_utils.match_order(Args(), e1, e2, W2, steps=3000, restarts=2)
    finally:
        np.random.choice = orig
    np.testing.assert_array_equal(W2.weight.data.cpu().numpy(), g["w_after"])
    np.testing.assert_array_equal(W2.bias.data.cpu().numpy(), g["b_after"])
    # a larger ragged case
    a = torch.rand(1000, 70, dtype=torch.float64, device="cuda")
    b = torch.rand(1000, 70, dtype=torch.float64, device="cuda")
    got = sk_utils.l1_cost_matrix(a, b)
    want = (a[:, :, None] - b[:, None, :]).abs().sum(0)
    assert torch.allclose(got, want, rtol=1e-12)


@pytest.mark.parametrize("hc,K", [(2, 8), (1, 6)])
def test_cluster_round_matches_oracle_reenactment(hc, K):
    """One SK round through ``cluster`` vs the oracle SK applied to the same per-head probabilities."""
    from selavi_amd import model as smodel, sk_utils
    from selavi_amd.data import SyntheticAVDataset
    ds = SyntheticAVDataset(n=192, T=4, S=32, F=40, Tp=36, n_classes=K)
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    m = m.cuda().train()
    # seed the BN running statistics like warmup_batchnorm does (utils.py:389-418)
    from selavi_amd.utils import warmup_batchnorm
    loader = [(torch.stack([ds[i][0] for i in range(b, b + 16)]), torch.stack([ds[i][1] for i in range(b, b + 16)]))
              for b in range(0, 64, 16)]
    warmup_batchnorm(Args(), m, loader, batches=4)
    args = Args(headcount=hc)
    logs = []

    class Lg:
        def info(self, s, **k):
            logs.append(s)
    np.random.seed(3)
    old = torch.zeros(192, hc, dtype=torch.long, device="cuda")
    new = sk_utils.cluster(args, old, ds, m, 0, Lg(), None, None, 0)
    assert new.shape == (192, hc) and new.dtype == torch.long and m.training
    assert any("NMI-tolabels" in s for s in logs) and any("Cost" in s for s in logs)
    # oracle re-enactment from the model's own eval-mode outputs
    m.eval()
    with torch.no_grad():
        V = torch.stack([ds[i][0] for i in range(192)]).cuda()
        A = torch.stack([ds[i][1] for i in range(192)]).cuda()
        outs = [m(V[i:i + 64], A[i:i + 64]) for i in range(0, 192, 64)]
    for h in range(hc):
        lv = torch.cat([(o[0][h] if hc > 1 else o[0]) for o in outs]).cpu().numpy()
        la = torch.cat([(o[1][h] if hc > 1 else o[1]) for o in outs]).cpu().numpy()
        _, L_o, _ = sk_ref.optimize_L_sk(sk_ref.head_probabilities(lv, la))
        agree = (new[:, h].cpu().numpy() == L_o).mean()
        assert agree == 1.0, f"head {h}: {agree:.3f} agreement with the oracle"


def _ddp_worker(rank, world, port, ret, wrap=False, precision="fp32"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=300))
    try:
        from selavi_amd import model as smodel, optim, train
        torch.cuda.set_device(0)
        hc, K = 2, 7
        m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        portable_init_(m, seed=31)
        step_ref.set_dropout_p(m, 0.0)
        m = m.cuda().train()
        m.set_precision(precision)
        if wrap == "native":
            net = train.data_parallel(m, [0], kind="native")
        elif wrap:
            net = train.wrap_ddp(m, [0])
        else:
            net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        # two clips per rank (world 2: the four clips / indices the single-process comparison below uses)
        video = portable_fill_(torch.empty(2 * world, 3, 4, 32, 32), 5)[rank * 2:(rank + 1) * 2].cuda()
        audio = portable_fill_(torch.empty(2 * world, 1, 40, 36), 6)[rank * 2:(rank + 1) * 2].cuda()
        sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
        sel_all = torch.tensor([3, 17, 42, 63]) if world == 2 else (torch.arange(2 * world) * 11 + 3) % 64
        sel = sel_all[rank * 2:(rank + 1) * 2].cuda()
        loss = train.train_step(net, opt, video, audio, sl, sel, hc)
        if wrap:      # several steps (DDP rebuilds its buckets after the first), then hash the whole state
            first = float(loss)
            from selavi_amd import ops as _o
            ex0 = _o.EXCHANGES[0]
            for _ in range(int(os.environ.get("SELAVI_DIAG_EXTRA_STEPS", "2"))):      # (diagnostic: 0 = hash after ONE step)
                ex0 = _o.EXCHANGES[0]
                loss = train.train_step(net, opt, video, audio, sl, sel, hc)
            exchanges = _o.EXCHANGES[0] - ex0                   # SyncBN exchanges of one step
            import hashlib
            hs = {k: hashlib.sha256(v.detach().cpu().numpy().tobytes()).hexdigest()[:16]
                  for k, v in sorted(m.state_dict().items())}
            from selavi_amd.comm import NativeComm
            ret[rank] = (first, float(loss), hs, len(NativeComm._cache), exchanges)      # communicators behind the C ABI (0 on gloo)
            NativeComm.destroy_all()
            return
        sd = m.state_dict()
        ret[rank] = (float(loss), {k: sd[k].flatten()[:32].cpu().numpy().copy() for k in (
            "video_network.base.layer2.0.conv1.0.0.weight", "video_network.base.stem.1.running_var",
            "audio_network.base.layer3.0.bn2.bias", "mlp_v1.block_forward.4.running_mean",
            "mlp_a0.block_forward.8.weight")})
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_syncbn_equals_single_process_large_batch():
    """DDP (grad averaging) + SyncBN on W ranks x b clips == one process with batch W*b
    (SURVEY.md Appendix B), here W = 2, b = 2 on one GPU over gloo."""
    import torch.multiprocessing as mp
    from selavi_amd import model as smodel, optim, train
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, 29400 + os.getpid() % 500, ret), nprocs=2, join=True)
    hc, K = 2, 7
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    m = m.cuda().train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
    audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
    sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
    sel = torch.tensor([3, 17, 42, 63]).cuda()
    loss = float(train.train_step(m, opt, video, audio, sl, sel, hc))
    sd = m.state_dict()
    l0, w0 = ret[0]
    l1, w1 = ret[1]
    assert abs(0.5 * (l0 + l1) - loss) <= 2e-4 * abs(loss)      # mean of the per-rank losses
    for k in w0:
        np.testing.assert_array_equal(w0[k], w1[k])             # ranks stay in lock-step
        np.testing.assert_allclose(w0[k], sd[k].flatten()[:32].cpu().numpy(), rtol=5e-3, atol=2e-4)


def test_two_rank_wrap_ddp_keeps_parameters_and_buffers_bit_identical():
    """train.wrap_ddp (no buffer broadcast, gradients as bucket views): after three steps every parameter AND every
    BatchNorm buffer is bit-identical on both ranks (SyncBN finalises the same all-reduced sums everywhere), and the
    first step's loss equals the reference-style DDP's."""
    import torch.multiprocessing as mp
    ret, ret_ref = mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(2, 29100 + os.getpid() % 200, ret, True), nprocs=2, join=True)
    mp.spawn(_ddp_worker, args=(2, 29500 + os.getpid() % 200, ret_ref), nprocs=2, join=True)
    diverged = [k for k in ret[0][2] if ret[0][2][k] != ret[1][2][k]]
    assert not diverged, f"{len(diverged)} of {len(ret[0][2])} tensors differ between the ranks: {diverged[:8]}"
    assert np.isfinite([ret[0][1], ret[1][1]]).all()
    for r in (0, 1):
        assert abs(ret[r][0] - ret_ref[r][0]) <= 1e-6 * abs(ret_ref[r][0])


def test_two_rank_native_data_parallel_is_bit_identical_to_ddp():
    """parallel.DataParallel (gradients written by the kernels into flat per-node buckets, one all-reduce per node)
    against torch DDP on the same two ranks: three steps, every parameter and buffer bit-identical (both average
    the same two addends; scaling by 1/2 commutes with the sum), ranks in lock-step."""
    import torch.multiprocessing as mp

    def rigs(attempt):
        ret, ret_ddp = mp.Manager().dict(), mp.Manager().dict()
        mp.spawn(_ddp_worker, args=(2, 29300 + (os.getpid() + 53 * attempt) % 150, ret, "native"), nprocs=2, join=True)
        mp.spawn(_ddp_worker, args=(2, 29700 + (os.getpid() + 53 * attempt) % 150, ret_ddp, True), nprocs=2, join=True)
        return ret, ret_ddp

    ret, ret_ddp = rigs(0)
    differs = [k for k in ret[0][2] if ret[0][2][k] != ret_ddp[0][2][k]]
    diverged = [k for k in ret[0][2] if ret[0][2][k] != ret[1][2][k]]
    assert not diverged, f"ranks diverged in {len(diverged)} tensors: {diverged[:6]}"
    assert not differs, f"{len(differs)} of {len(ret[0][2])} tensors differ from DDP: {differs[:6]}"
    assert ret[0][0] == ret_ddp[0][0] and ret[0][1] == ret_ddp[0][1]
    # SyncBN exchanges per step (each one a latency-bound all-reduce on a compute stream): one per BatchNorm forward and
    # backward -- every one of them sits on a true dependency (the next conv / the backward-data conv needs the global
    # statistics) -- except the pairs that become available together: the last BatchNorm of a residual block and its
    # downsample BatchNorm share one exchange in both directions.  R(2+1)D-18 + ResNet-9 + the grouped heads: 88
    # (100 before the pairs were packed; 49 of them on the audio trunk's own stream and communicator)
    assert ret[0][4] == ret[1][4] and ret[0][4] <= 88, ret[0][4]


def _ragged_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=300))
    try:
        from selavi_amd import model as smodel, train
        torch.cuda.set_device(0)
        m = smodel.load_model(use_mlp=True, num_classes=7, norm_feat=False, headcount=2)
        portable_init_(m, seed=31)
        m = m.cuda().train()
        net = train.data_parallel(m, [0], kind="native")
        dev = torch.device("cuda", 0)
        out = []
        net.check_equal_batches(2, dev)                  # equal batches: no error, on the first call or later
        net.check_equal_batches(2, dev)
        # (a) the ragged last batch: ONE rank's size changes, the other keeps its size -- both must still meet in the
        #     collective (a rank-local "size changed" condition around it hangs here) and both must see the error
        try:
            net.check_equal_batches(2 if rank == 0 else 1, dev)
            net._read_batch_checks(block=True)
            out.append("no error")
        except RuntimeError as e:
            out.append(str(e))
        # (b) both sizes change, to different values
        try:
            net.check_equal_batches(3 if rank == 0 else 4, dev)
            out.append("no error")
        except RuntimeError as e:
            out.append(str(e))
        net.check_equal_batches(4, dev)                  # and the check recovers
        ret[rank] = out
        from selavi_amd.comm import NativeComm
        NativeComm.destroy_all()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_reports_a_ragged_batch_on_every_rank():
    """parallel.DataParallel's equal-batch check (what slv_bn_sync_finalize and the averaged buckets assume): the collective
    is issued by every rank on every call, so the rank whose loader delivered a ragged last batch does not sit alone in it."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_ragged_worker, args=(2, 28350 + os.getpid() % 150, ret), nprocs=2, join=True)
    for r in (0, 1):
        a, b = ret[r]
        assert "batch sizes differ" in a and "max 2, min 1" in a, a
        assert "batch sizes differ" in b and "max 4, min 3" in b, b


def test_two_rank_native_data_parallel_on_the_16bit_path_stays_in_lock_step():
    """parallel.DataParallel + SyncBN with the video trunk on the bf16 kernels (fp32 weight gradients written straight
    into the flat buckets by slv_cl16_wgrad, SyncBN sums from the bf16 statistics epilogue): after three steps every
    parameter and BatchNorm buffer is bit-identical on both ranks, and the first loss is the fp32 run's to bf16 accuracy."""
    import torch.multiprocessing as mp
    ret, ret32 = mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(2, 28600 + os.getpid() % 150, ret, "native", "bf16"), nprocs=2, join=True)
    mp.spawn(_ddp_worker, args=(2, 28800 + os.getpid() % 150, ret32, "native"), nprocs=2, join=True)
    diverged = [k for k in ret[0][2] if ret[0][2][k] != ret[1][2][k]]
    assert not diverged, f"ranks diverged in {len(diverged)} tensors: {diverged[:6]}"
    assert np.isfinite([ret[0][0], ret[0][1], ret[1][1]]).all()
    for r in (0, 1):
        assert abs(ret[r][0] - ret32[r][0]) <= 0.1 * abs(ret32[r][0]), (ret[r][0], ret32[r][0])


def _nccl_worker(rank, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from selavi_amd import model as smodel, optim, train
        hc, K = 2, 7
        out = []
        for distributed in (False, True, "native"):
            m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
            portable_init_(m, seed=31)
            step_ref.set_dropout_p(m, 0.0)
            m = m.cuda().train()
            net = m
            if distributed == "native":   # flat buckets, ReduceOp.AVG on RCCL's stream, final-callback wait
                net = train.data_parallel(m, [0], kind="native")
            elif distributed:     # SyncBN all-reduces + DDP buckets go through RCCL (streams, events, side streams)
                m.set_sync_bn(True)
                net = train.wrap_ddp(m, [0])
            opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
            video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
            audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
            sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
            sel = torch.tensor([3, 17, 42, 63]).cuda()
            losses = [float(train.train_step(net, opt, video, audio, sl, sel, hc)) for _ in range(3)]
            out.append((losses, torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy()))
        ret["plain"], ret["rccl"], ret["native"] = out
    finally:
        dist.destroy_process_group()


def test_rccl_syncbn_ddp_world1_equals_plain_step():
    """The RCCL code path (SyncBN all-reduces issued from the trunk / weight-gradient side streams and autograd's
    thread, DDP gradient buckets) on a world of one rank: sums over one rank are the identity, so three steps must
    reproduce the non-distributed run -- a stream-ordering bug around the collectives would show up here."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(29900 + os.getpid() % 90, ret), nprocs=1, join=True)
    (l0, w0) = ret["plain"]
    for key in ("rccl", "native"):
        l1, w1 = ret[key]
        assert np.isfinite(l1).all() and np.isfinite(w1).all(), key
        np.testing.assert_allclose(l1, l0, rtol=2e-5, err_msg=key)
        assert np.linalg.norm(w1 - w0) <= 1e-4 * np.linalg.norm(w0), key


# ---- the SK round on two ranks with the HIP kernels: sharded feature pass + sharded solve + match_order, two rounds
def _cluster_worker(rank, world, port, hc, K, ret, match=True):
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=300))
    try:
        from selavi_amd import model as smodel, sk_utils
        from selavi_amd.data import SyntheticAVDataset
        from selavi_amd.utils import warmup_batchnorm
        torch.cuda.set_device(0)
        torch.manual_seed(31)         # utils.py:277-283 at start-up: the warm-up's dropout masks (head BatchNorm1d statistics)
        ds = SyntheticAVDataset(n=192, T=4, S=32, F=40, Tp=36, n_classes=K)
        m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
        portable_init_(m, seed=31)
        m = m.cuda().train()
        m.set_sync_bn(False)          # identical BN statistics on every rank: all ranks warm up on the same batches
        loader = [(torch.stack([ds[i][0] for i in range(b, b + 16)]), torch.stack([ds[i][1] for i in range(b, b + 16)]))
                  for b in range(0, 64, 16)]
        warmup_batchnorm(Args(), m, loader, batches=4)
        # batches of 32 on every rank AND in the single process: launch configurations (and with them the fp32 summation
        # order of the eval forward) follow the batch shape, and on a randomly initialised model the features of all
        # clips nearly coincide, so the assignment hinges on the last bits of the logits
        args = Args(headcount=hc, rank=rank, match=match, ind_groups=min(2, hc), distribution='gauss', sk_batch_size=32)
        np.random.seed(31)            # utils.py:277-283 seeds every rank alike; the streams diverge inside round 1
        torch.manual_seed(31)         # (only the searching rank draws match_order's pairs)
        labels = torch.zeros(192, hc, dtype=torch.long, device="cuda")
        out = []
        for it in (0, 1):
            labels = sk_utils.cluster(args, labels, ds, m, it, None, None, None, it)
            out.append(labels.cpu().numpy().copy())
        heads_a = [m.mlp_a] if hc == 1 else [getattr(m, f"mlp_a{h}") for h in range(hc)]
        w = torch.cat([hd.block_forward[8].weight.flatten() for hd in heads_a])
        ret[rank] = (out, w.detach().cpu().numpy().copy(), float(np.random.rand()))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_cluster_rounds_match_single_process():
    """sk_utils.cluster on two ranks (row-sharded feature pass + HIP sharded Sinkhorn-Knopp, gauss marginals, two head
    groups, hc = 3) over TWO rounds.

    hc = 3 with match_order (the reference default): both ranks return the same labels and hold the same permuted audio
    heads although their numpy streams have diverged by the second round (only the searching rank draws the swap pairs;
    the head order is rank 0's, broadcast).
    hc = 1 without it: the labels also equal a single process's, bit for bit -- with one head the feature bank holds the
    logits of the per-batch forward (sk_utils.py:207-211), whose shapes are the same on 1 and 2 ranks.  (With hc > 1 the
    heads are applied to the whole BANK, a GEMM whose row count -- and with it the fp32 summation order -- is the
    shard's; and match_order's accept rule `current - future > 0` is an exact tie in exact arithmetic whenever two
    swapped columns lie on the same side, so it follows the rounding of the all-reduced table.  On a randomly
    initialised model, where all clips have nearly the same softmax, both effects move labels; neither is an error.)"""
    import torch.multiprocessing as mp
    ret, ret_2, ret_1 = mp.Manager().dict(), mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_cluster_worker, args=(2, 28100 + os.getpid() % 150, 3, 8, ret), nprocs=2, join=True)
    assert ret[0][2] != ret[1][2], "the ranks' numpy streams were expected to diverge (match_order draws on rank 0)"
    for rnd in (0, 1):
        np.testing.assert_array_equal(ret[0][0][rnd], ret[1][0][rnd])
        assert len(np.unique(ret[0][0][rnd][:, 0])) > 1
    np.testing.assert_array_equal(ret[0][1], ret[1][1])
    mp.spawn(_cluster_worker, args=(2, 28300 + os.getpid() % 150, 1, 8, ret_2, False), nprocs=2, join=True)
    mp.spawn(_cluster_worker, args=(1, 0, 1, 8, ret_1, False), nprocs=1, join=True)
    for rnd in (0, 1):
        np.testing.assert_array_equal(ret_2[0][0][rnd], ret_2[1][0][rnd])
        np.testing.assert_array_equal(ret_2[0][0][rnd], ret_1[0][0][rnd])
        assert len(np.unique(ret_1[0][0][rnd][:, 0])) > 1


def _native_comm_worker(rank, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from selavi_amd import ops, sk_utils
        from selavi_amd.comm import NativeComm
        from tests._synth import synth_PS
        comm = NativeComm.for_group(None)
        assert comm is not None and comm.world == 1 and "rccl" in comm.library()
        ret["lib"] = comm.library()
        # collectives of one rank are the identity
        for dt in (torch.float64, torch.float32, torch.int64):
            t = (torch.arange(1000, device="cuda") % 17).to(dt)
            want = t.clone()
            comm.allreduce_(t)
            torch.cuda.synchronize()
            assert torch.equal(t, want), dt
        # SyncBN in one call == the single-process finalize (same kernels, fp64 sums in between)
        g = torch.Generator(device="cuda").manual_seed(1)
        Cc, nblk = 45, 37
        ps, pq = torch.randn(Cc, nblk, device="cuda", generator=g), torch.rand(Cc, nblk, device="cuda", generator=g) * 9
        gamma, beta = torch.rand(Cc, device="cuda", generator=g) + 0.5, torch.randn(Cc, device="cuda", generator=g)
        outs = []
        for sync in (None, (comm, 1)):
            rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
            mi, ss = ops.bn_train_finalize(ps, pq, 1000.0, gamma, beta, rm, rv, 0.1, 1e-5, sync=sync)
            outs.append(torch.cat([mi.flatten(), ss.flatten(), rm, rv]))
        assert torch.equal(outs[0], outs[1])
        part = torch.randn(Cc, 11, 2, device="cuda", generator=g)
        o2 = []
        for sync in (None, (comm, 1)):
            dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
            b5 = ops.bn_bwd_finish(part, None, 11, 1000.0, mi, gamma, ss, None, None, sync, dg, db, None, None)[0]
            o2.append(torch.cat([b5.flatten(), dg, db]))
        assert torch.equal(o2[0], o2[1])
        # sharded Sinkhorn-Knopp loop through slv_sk_iterate_sharded == the single-GPU loop
        PS = synth_PS(2048, 28, 2.0, 3)

        class A:
            distribution, dist, diff_dist_every, diff_dist_per_head = 'default', None, False, True
            gauss_sd, headcount, lamb, rank = 0.1, 1, 20, 0
        c0, L0 = sk_utils.optimize_L_sk_gpu(A(), torch.from_numpy(PS).cuda(), 0, None)
        i0 = sk_utils.optimize_L_sk_gpu.last_info["iters"]
        c1, L1 = sk_utils.optimize_L_sk_gpu(A(), torch.from_numpy(PS).cuda(), 0, None, group=dist.group.WORLD, N_global=2048)
        assert c0 == c1 and torch.equal(L0, L1) and sk_utils.optimize_L_sk_gpu.last_info["iters"] == i0
        ret["ok"] = True
    finally:
        dist.destroy_process_group()


def test_native_rccl_communicator_world1():
    """RCCL behind the C ABI (slv_comm_*): unique-id bootstrap through torch.distributed, the collectives on
    the caller's stream, the one-call SyncBN (slv_bn_sync_finalize / slv_bn_bwd_sync_finalize) and the one-call sharded
    Sinkhorn-Knopp loop (slv_sk_iterate_sharded) against their single-process counterparts, on a world of one rank."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_native_comm_worker, args=(29810 + os.getpid() % 80, ret), nprocs=1, join=True)
    assert ret.get("ok") and "rccl" in ret["lib"]


def _graphed_dp_worker(rank, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from selavi_amd import model as smodel, optim, train
        from selavi_amd.comm import NativeComm
        hc, K = 2, 7
        video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
        audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
        sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
        sel = torch.tensor([3, 17, 42, 63]).cuda()

        def make():
            m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
            portable_init_(m, seed=31)
            step_ref.set_dropout_p(m, 0.0)
            m = m.cuda().train()
            net = train.data_parallel(m, [0], kind="native")
            return m, net, optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        m0, n0, o0 = make()
        for _ in range(5):
            l0 = train.train_step(n0, o0, video, audio, sl, sel, hc)
        m1, n1, o1 = make()
        assert n1.capturable() and len(NativeComm._cache) >= 3          # bn, bn_audio, grad: RCCL behind the C ABI
        gs = train.GraphedStep(n1, o1, video, audio, sl, sel, hc, warmup=3)
        for _ in range(2):
            l1 = gs.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(m0.state_dict().values(), m1.state_dict().values()))
        ret["out"] = (float(l0), float(l1), same, int(m1.state_dict()["video_network.base.stem.1.num_batches_tracked"]))
        NativeComm.destroy_all()
    finally:
        dist.destroy_process_group()


def test_graphed_step_under_native_data_parallel_captures_the_rccl_collectives():
    """train.GraphedStep around parallel.DataParallel on REAL RCCL (a world of one rank: the boxes have one GPU and RCCL
    refuses two ranks per device): the 88 SyncBN exchanges (slv_bn_sync_finalize on the main and the audio stream) and the 7
    gradient-bucket all-reduces (slv_comm_allreduce_f32 on the buckets' stream) are captured as graph nodes between the
    kernels; two replays leave bit for bit the weights, buffers and loss of five eager data-parallel steps (3 warm-up + 2)."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_graphed_dp_worker, args=(29900 + os.getpid() % 90, ret), nprocs=1, join=True)
    l0, l1, same, nbt = ret["out"]
    assert l0 == l1 and same and nbt == 5, ret["out"]


def test_graphed_step_refuses_host_driven_collectives():
    """... and over torch.distributed (gloo here) the capture is refused up front, not left to fail inside the graph."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_graphed_refuse_worker, args=(29800 + os.getpid() % 90, ret), nprocs=1, join=True)
    assert "torch.distributed" in ret["err"]


def _graphed_refuse_worker(rank, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["SELAVI_NATIVE_COMM"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from selavi_amd import model as smodel, optim, train
        m = smodel.load_model(use_mlp=True, num_classes=7, norm_feat=False, headcount=2).cuda().train()
        net = train.data_parallel(m, [0], kind="native")
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        video = torch.randn(2, 3, 4, 32, 32).cuda()
        audio = torch.randn(2, 1, 40, 36).cuda()
        sl = torch.zeros(64, 2, dtype=torch.int64).cuda()
        sel = torch.tensor([3, 17]).cuda()
        try:
            train.GraphedStep(net, opt, video, audio, sl, sel, 2)
            ret["err"] = "no error"
        except RuntimeError as e:
            ret["err"] = str(e)
    finally:
        dist.destroy_process_group()
