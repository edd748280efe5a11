#!/usr/bin/env python
"""bench.py -- SeLaVi training-step throughput on N MI355X of one node (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch per GPU: R(2+1)D-18 video + ResNet-9
audio forward, 2*hc MLP heads, cross-entropy on the SK pseudo labels, backward, fused SGD
(/root/reference/main.py:284-302).  Workload = BASELINE.json configs[1] ("cfg2"): per-GPU batch 16,
16x112x112 clips, 1x129x100 log-mel, K=309, headcount 10, fp32 (weak scaling: cfg3 at N=8).
The SK solver is timed separately (iterations/s at the VGG-Sound size N=170752, K=309, row-sharded
over the N ranks) and reported in the same JSON line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(batch=16, T=16, S=112, F=129, Tp=100, K=309, hc=10, N=170752)
# algorithmic work per clip of the forward (fused BN/ReLU): video 81.04 + audio 0.506 GFLOP
# + heads 0.0168 GFLOP; step = 3x forward (dgrad + wgrad)
FWD_GFLOP_PER_CLIP = 81.04 + 0.506 + 0.0168
PEAK_FP32_MFMA_TF = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA == fp32 vector peak
# the fp32 convs as they run by default (csrc/igemm3.hpp): every fp32 operand cut exactly into 3 bf16 pieces, a product =
# 6 partial products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -> the binding roofline of an fp32 conv FLOP is the
# dense bf16 MFMA peak / 6
PEAK_BF16_MFMA_TF_DENSE = 2500.0
PEAK_X3_TF = PEAK_BF16_MFMA_TF_DENSE / 6.0
PEAK_HBM_GBS = 8000.0


FWD_MB_PER_CLIP = 518.1 + 3.38          # sum over convs of (in + out) fp32 bytes, BN/ReLU/add/pool fused


def _pmc_traffic(key):
    """Measured HBM bytes/launch recorded by the PMC passes of this round (tools/pmc_traffic.sh ->
    profiles/r01_pmc.json; None if not recorded)."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            v = d.get(key)
            if isinstance(v, dict) and v.get("hbm_bytes_per_launch"):
                return v.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
    return None


def _library_digest():
    try:
        return open(os.path.join(ROOT, "selavi_amd", "libselavi_hip.so.stamp")).read().strip()
    except OSError:
        return None


def _rocprof_in_step(kernel, grid, files):
    """Average duration (ms) of `kernel [grid]` INSIDE the training step, from a committed rocprofv3 --kernel-trace
    summary of this very command (tools/prof_r5.sh -> tools/rocprof_summary.py).  A FROZEN figure (the trace was taken on
    another box of the pool, +-3 % run to run; rocprofv3 cannot wrap the bench from inside), reported only when the summary
    carries the digest of the library this run loaded ("library digest ..." line, = selavi_amd/build.py:_digest()): a summary
    of other kernels would go stale silently.  The LIVE in-step figure of the line is roofline.in_step_live (HIP events
    around the launch inside the step); None if no matching summary is committed."""
    import re
    dig = _library_digest()
    for name in files:
        try:
            lines = open(os.path.join(ROOT, "profiles", name)).read().splitlines()
        except OSError:
            continue
        m0 = re.match(r"library digest (\S+)", lines[0]) if lines else None
        if not (m0 and dig and m0.group(1) == dig):
            continue
        for line in lines:
            m = re.match(r"\s*[\d.]+\s+[\d.]+\s+(\d+)\s+([\d.]+)\s+\d+\s+\d+\s+\d+\s+(.*?)\s*\[(\d+)\]\s*$", line)
            if m and m.group(3).startswith(kernel) and int(m.group(4)) == grid:
                return dict(ms=float(m.group(2)) / 1e3, launches=int(m.group(1)), source="profiles/" + name,
                            library_digest=dig[:16],
                            frozen="committed rocprofv3 trace of this command on the same library build, not measured by this run")
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=CFG2["batch"], help="per-GPU batch (cfg2: 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sk", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=CFG2["batch"], help="batch of the CPU baseline (cfg2: 16, the GPU step's)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the 16-bit leg (BASELINE configs[4])")
    ap.add_argument("--no-native-leg", action="store_true", help="skip the comparison leg on the native fp32 MFMA kernels")
    ap.add_argument("--cfg5-batch", type=int, default=CFG5["batch"], help="per-GPU batch of the 16-bit leg (cfg5: 128)")
    ap.add_argument("--cfg5-steps", type=int, default=15)
    ap.add_argument("--cfg5-warmup", type=int, default=5)
    return ap.parse_args()


def hot_conv_roofline(batch, dev):
    """Dominant kernel: the implicit-GEMM conv (csrc/igemm.hpp).  Its heaviest single launch in the
    step is the layer-1 spatial conv Conv3d(64->144,(1,3,3)) on B x 64 x 16 x 56 x 56 (4 forward
    launches/step + the matching dgrad/wgrad).  Time that exact launch with HIP events on the
    stream it runs on; achieved = 2*MACs/launch / average duration."""
    from selavi_amd import ops
    shape = (batch, 64, 16, 56, 56)
    plan = ops.ConvPlan.get(shape, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), dev)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(*shape, device=dev, generator=g)
    w = torch.randn(144, 64, 1, 3, 3, device=dev, generator=g) * 0.04
    ss = torch.stack([torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1])
    wf, _ = ops.conv_w_transform(plan, w, need_wt=False)      # per-step weight re-layout, outside the launch timed here
    for _ in range(3):
        ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * batch * 16 * 56 * 56 * 144 * 64 * 9
    cfg = plan.cfg_fwd
    tile = "MT=%d,NT=%d,K-slices=%d" % (cfg & 255, (cfg >> 8) & 15, cfg >> 16) if cfg else "heuristic tile (MT=9,NT=2)"
    x3 = ops.conv_arithmetic() == "x3"
    name = ("igemm3_kernel<%s, BN+ReLU prologue> (fp32 tensors, operands split into 3 bf16 pieces in registers, 6 x "
            "v_mfma_f32_16x16x32_bf16 per product tile, fp32 accumulate)" if x3 else
            "igemm_kernel<MODE_CONV, %s, BN+ReLU prologue, tap-major K> (v_mfma_f32_16x16x4_f32)") % tile
    return dict(kernel=name + " layer1 (1,3,3) 64->144 forward", ms=ms, flop=flop, tflops=flop / ms / 1e9, x3=x3,
                grid=plan.nblk * ((144 + (cfg & 255 or 9) * 16 - 1) // ((cfg & 255 or 9) * 16)))


def sk_bench(rank, world, dev, iters=50):
    """SK iterations/s at the VGG-Sound size, rows sharded over the ranks."""
    import torch.distributed as dist
    from selavi_amd import sk_utils
    N, K = CFG2["N"], CFG2["K"]
    lo, hi = rank * N // world, (rank + 1) * N // world
    n = hi - lo
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    lv = torch.randn(n, K, device=dev, generator=g)
    la = torch.randn(n, K, device=dev, generator=g)
    P = sk_utils.head_probabilities(lv, la, power=10.0)
    be = sk_utils._HIP
    grid = be.default_grid(n, K)
    ws = be.workspace(K, grid, dev)
    beta = torch.empty(n, dtype=torch.float64, device=dev)
    r = torch.full((K,), 1.0 / K, dtype=torch.float64, device=dev)
    grp = dist.group.WORLD if world > 1 else None
    comm = sk_utils._comm_of(grp, be)        # RCCL behind the C ABI (None over gloo: torch.distributed carries it)

    TOL = -1.0      # err >= 0 > tol: the loop never declares itself done (with tol = 0 it reaches an exact fixed point, err == 0.0,
                    # after a few hundred iterations and every later launch is a no-op: the timed ones are checked below)

    def run(k):
        if world == 1:
            be.iterate(P, beta, r, TOL, 10 ** 9, k, ws, grid)
        elif comm is not None:               # the product's sharded loop: k iterations in one host call, one stream
            from selavi_amd._lib import C, ptr, stream
            C.slv_sk_iterate_sharded(comm.h, ptr(P), n, N, K, ptr(beta), ptr(r), TOL, 10 ** 9, k, ptr(ws), grid, stream())
        else:
            sv = be.s_view(ws, K, grid)
            for _ in range(k):
                be.pass_reduce(P, N, beta, ws, grid)
                dist.all_reduce(sv, group=grp)
                be.update(r, K, TOL, 10 ** 9, False, ws, grid)
    be.begin(P, N, beta, ws, grid)
    be.local_reduce(K, ws, grid)
    if world > 1:
        sk_utils._allreduce(be.s_view(ws, K, grid), grp, comm)
    be.update(r, K, TOL, 10 ** 9, True, ws, grid)
    run(5)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(iters)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    hb = torch.empty(4, dtype=torch.float64, pin_memory=True)
    be.status_async(ws, K, grid, hb).synchronize()
    assert int(hb[0]) == 5 + iters and int(hb[1]) == 0, f"the SK loop stopped early: counter {hb[0]}, done {hb[1]}"
    if world > 1:
        t = torch.tensor([ms], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = t.item()
    split = None
    if world > 1:
        # the same shard without the exchange (the single-GPU loop on the local rows): what of an iteration is the pass over
        # the shard and what is the K-vector all-reduce (+ its launch latency) -- SURVEY 8e-2
        be.iterate(P, beta, r, TOL, 10 ** 9, 5, ws, grid)
        torch.cuda.synchronize()
        e0.record()
        be.iterate(P, beta, r, TOL, 10 ** 9, iters, ws, grid)
        e1.record()
        torch.cuda.synchronize()
        ms_local = e0.elapsed_time(e1) / iters
        t = torch.tensor([ms_local], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_local = t.item()
        split = dict(us_pass_reduce_update=ms_local * 1e3, us_allreduce=max(ms - ms_local, 0.0) * 1e3,
                     transport="native: slv_sk_iterate_sharded (RCCL on the compute stream)" if comm is not None
                     else "torch.distributed all_reduce per iteration")
    gbs = n * K * 8 / ms / 1e6      # per-GPU algorithmic bytes (one read of the fp64 shard) / time
    note = None if world == 1 else ("rows sharded over %d GPUs: the pass over a shard takes ~%.0f us, each iteration is then bound by "
                                    "the latency of its K-vector all-reduce (RCCL on the compute stream, slv_sk_iterate_sharded)" % (world, 76.0 / world))
    return dict(iters_per_s=1e3 / ms, us_per_iter=ms * 1e3, us_per_iter_split=split, N=N, K=K, rows_per_gpu=n, grid=grid, note=note,
                roofline=dict(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS,
                              traffic=_pmc_traffic("sk_pass") if world == 1 else None))


def sk_round_estimate(m, dev, world, step_clips_per_s, sk):
    """What one Sinkhorn-Knopp round costs next to the training it interleaves with (BASELINE metric: "clips/sec
    (video+audio fwd/bwd+SK)").  Measured here: the eval-mode feature pass of sk_utils.py:137-233 at its batch size
    (64, :168) on this GPU, as selavi_amd.sk_utils runs it by default: the model's own fp32 eval forward.  Beside it: the same
    trunks with BatchNorm folded into the weights for the length of the pass (selavi_amd/infer32.py) with the exact three-piece
    operand split ("fp32_folded") and with two pieces per operand ("fp32x2", opt-in), and the bf16 opt-in.
    Derived with the reference's defaults (opt.py:71,88,102: 100 epochs, nopts=100 rounds,
    ind_groups=1) at the VGG-Sound size: round = N / (W x feature-pass rate) + hc heads x 200 SK iterations at the
    measured it/s (71-271 iterations to converge), against the epochs x N / nopts clips trained between
    two rounds."""
    from selavi_amd import infer32
    B = 64
    g = torch.Generator(device=dev).manual_seed(77)
    video = torch.randn(B, 3, CFG2["T"], CFG2["S"], CFG2["S"], device=dev, generator=g)
    audio = torch.randn(B, 1, CFG2["F"], CFG2["Tp"], device=dev, generator=g)
    m.eval()
    m.return_features = True

    def rate_of(fn, reps=4):
        fn()
        fn()                                   # (plans, launch configurations, the folded weight images)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return B / ((time.perf_counter() - t0) / reps)
    rates = {}
    try:
        with torch.no_grad():
            rates["fp32"] = rate_of(lambda: m(video, audio))
            for name, pieces in (("fp32_folded", 3), ("fp32x2", 2)):
                try:
                    with infer32.folded_eval(m, pieces=pieces):
                        rates[name] = rate_of(lambda: m(video, audio))
                except Exception as e:             # never take the bench line down with a side measurement
                    print(f"feature pass {name} not measured: {e!r}", file=sys.stderr)
        # opt-in alternative (NOT the bit-exact path): the same pass in bf16 on the channels-last MFMA kernels
        try:
            from selavi_amd import infer16
            eng = infer16.Engine(m)
            rates["bf16"] = rate_of(lambda: eng.features(video, audio))
            del eng
        except Exception as e:                     # experimental path: never take the bench line down with it
            print(f"bf16 feature pass not measured: {e!r}", file=sys.stderr)
    finally:
        m.return_features = False
        m.train()
    N, hc, epochs, rounds, its = CFG2["N"], CFG2["hc"], 100, 100, 200
    t_sk = hc * its / sk["iters_per_s"] if sk else float("nan")
    t_train = epochs * N / rounds / step_clips_per_s

    def incl(rate):
        return step_clips_per_s * t_train / (N / (rate * world) + t_sk + t_train)
    default = "fp32"
    rate = rates[default]
    t_feat = N / (rate * world)
    notes = {"fp32": "default: the model's own eval forward (bit for bit what model.eval() returns)",
             "fp32_folded": "SELAVI_FEATURE_PASS=fp32_folded: BatchNorm folded into the weights, conv + BN (+ shortcut) + ReLU in one "
                            "launch, weight images made once per pass, exact three-piece operand split (features 6e-7 off the default)",
             "fp32x2": "SELAVI_FEATURE_PASS=fp32x2 (opt-in): folded, two bf16 pieces per operand / three partial products: features ~2e-4 "
                       "relative off the exact split, labels not guaranteed identical",
             "bf16": "SELAVI_FEATURE_PASS=bf16 (opt-in): eval forward on bf16 channels-last activations (selavi_amd/infer16.py); features "
                     "within ~5e-3 of fp32, labels not bit-exact"}
    return {"feature_pass": default, "feature_pass_clips_per_s_per_gpu": rate, "feature_pass_batch": B, "feature_pass_s": t_feat,
            "feature_pass_mfma_frac": rate * FWD_GFLOP_PER_CLIP / 1e3 / PEAK_X3_TF,
            "sk_solve_s": t_sk, "round_s": t_feat + t_sk, "training_between_rounds_s": t_train,
            "fraction_of_wall_clock": (t_feat + t_sk) / (t_feat + t_sk + t_train),
            "clips_per_s_including_sk": incl(rate),
            "assumes": f"N={N}, hc={hc}, {rounds} rounds over {epochs} epochs, ind_groups=1, {its} SK iterations per head",
            "by_feature_pass": {k: {"feature_pass_clips_per_s_per_gpu": v, "clips_per_s_including_sk": incl(v), "note": notes[k]}
                                for k, v in rates.items()},
            "bf16_feature_pass_opt_in": None if "bf16" not in rates else {
                "feature_pass_clips_per_s_per_gpu": rates["bf16"],
                "hbm_roofline_frac": rates["bf16"] * FWD_MB_PER_CLIP / 2 / 1e3 / PEAK_HBM_GBS,
                "clips_per_s_including_sk": incl(rates["bf16"]), "note": notes["bf16"]}}


CFG5 = dict(batch=128, T=32, S=112, F=129, Tp=100, K=309, hc=10)
FWD_GFLOP_PER_CLIP_T32 = 162.08 + 0.506 + 0.0168      # SURVEY 8d, T = 32
FWD_MB_PER_CLIP_T32_BF16 = 1036.2 / 2 + 3.38           # video bytes halve in bf16 (the audio trunk's 3.4 MB/clip are kept at the fp32 figure)
PEAK_BF16_MFMA_TF = 2500.0


HOT_X3_KERNEL = "igemm3_kernel<9, 2, 1, 0, 4, 2, true, 3>"     # csrc/igemm3.hpp: MT, NT, PRO, EPI, WAVES, OCC, FUSE, NP of the layer-1 spatial forward
RIDGE_BF16 = PEAK_BF16_MFMA_TF_DENSE * 1e3 / PEAK_HBM_GBS      # 312.5 FLOP/B: above it a bf16 kernel is MFMA-bound, below HBM-bound
# The two layer-1 forward kernels of the 16-bit path (68 % of the forward's bytes, 45 % of its FLOPs), one on each side of the ridge:
#   spatial  Conv3d(64 -> 144, (1,3,3)): 2*144*64*9 FLOP / ((64+144)*2 B) = 399 FLOP/B  -> MFMA-bound
#   temporal Conv3d(144 -> 64, (3,1,1)): 2*144*64*3 FLOP / ((144+64)*2 B) = 133 FLOP/B  -> HBM-bound   
HOT16 = {
    "l1_spatial": dict(cin=64, cout=144, k=(1, 3, 3), pad=(0, 1, 1), kernel="conv_cl16_sr_kernel<1, 1>", grid=256, pmc="hot_conv16_fwd",
                       what="conv_cl16_sr_kernel<1,1> (csrc/conv_cl16_sr.hip: weights resident in registers, three MFMA waves + a "
                            "data-movement wave per CU, persistent) layer1 (1,3,3) 64->144 train forward"),
    "l1_temporal": dict(cin=144, cout=64, k=(3, 1, 1), pad=(1, 0, 0), kernel="conv_cl16_tr_kernel<4, 5, 1, 1>", grid=1024,
                        pmc="hot_conv16_fwd_temporal",
                        what="conv_cl16_tr_kernel<4,5,1,1> (csrc/conv_cl16_tr.hip: weights resident in registers, 32-pixel columns "
                             "walked frame by frame) layer1 (3,1,1) 144->64 train forward"),
}


def _conv16_holder(L):
    class Conv:
        in_channels, out_channels, kernel3, stride3, padding3 = L["cin"], L["cout"], L["k"], (1, 1, 1), L["pad"]
    return Conv


def hot_conv16_roofline(name, batch, T, dev):
    """One of the two dominant kernels of the 16-bit forward (HOT16) in train mode (BatchNorm + ReLU prologue on load,
    statistics epilogue), alone on the chip, timed with HIP events on its stream.  Algorithmic bytes = input + output in bf16
    ((Cin + Cout) channels x 2 B per position), FLOPs = 2 Cin Cout taps per position."""
    from selavi_amd import ops16
    L = HOT16[name]
    g = torch.Generator(device=dev).manual_seed(7)
    x = ops16.to_channels_last16(torch.randn(batch, L["cin"], T, 56, 56, device=dev, generator=g))
    plan = ops16.plan_for(x, _conv16_holder(L))
    w = torch.randn(L["cout"], L["cin"], *L["k"], device=dev, generator=g) * 0.04
    ss = torch.stack([torch.rand(L["cin"], device=dev, generator=g) + 0.5, torch.randn(L["cin"], device=dev, generator=g) * 0.1]).contiguous()
    wf, _ = ops16.conv_w_transform(plan, w, need_wt=False)
    for _ in range(3):
        ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
    e1.record()
    torch.cuda.synchronize()
    return dict(ms=e0.elapsed_time(e1) / reps, clips=batch)


def _roof16(name, T, iso, live, frozen, pmc_bytes_16x16):
    """Roofline object of one HOT16 kernel: `bound` follows its arithmetic intensity against the ridge; `frac` is the fraction
    of THAT roof; the other roof's fraction rides along.  iso = isolated launch, live = the launches inside the cfg5 step
    (HIP events on the launch's stream, this run), frozen = the committed rocprofv3 average of the same launch (digest-matched)."""
    L = HOT16[name]
    taps = L["k"][0] * L["k"][1] * L["k"][2]
    flop_pos, byte_pos = 2.0 * L["cin"] * L["cout"] * taps, (L["cin"] + L["cout"]) * 2.0
    ai = flop_pos / byte_pos
    bound = "mfma" if ai > RIDGE_BF16 else "hbm"

    def rates(ms, clips):
        pos = clips * T * 56 * 56
        gbs, tf = pos * byte_pos / ms / 1e6, pos * flop_pos / ms / 1e9
        return dict(ms=ms, clips_per_launch=clips, hbm_gbs=gbs, hbm_frac=gbs / PEAK_HBM_GBS, mfma_tflops=tf,
                    mfma_frac=tf / PEAK_BF16_MFMA_TF_DENSE,
                    achieved=tf if bound == "mfma" else gbs, frac=tf / PEAK_BF16_MFMA_TF_DENSE if bound == "mfma" else gbs / PEAK_HBM_GBS)
    r = rates(iso["ms"], iso["clips"])
    out = {"bound": bound, "achieved": r["achieved"], "peak": PEAK_BF16_MFMA_TF_DENSE if bound == "mfma" else PEAK_HBM_GBS,
           "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": r["frac"],
           "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": RIDGE_BF16,
           "kernel": L["what"] + " at bs %d (alone on the chip)" % iso["clips"], "ms_per_launch": iso["ms"],
           "hbm_gbs": r["hbm_gbs"], "hbm_frac": r["hbm_frac"], "mfma_tflops": r["mfma_tflops"], "mfma_frac": r["mfma_frac"],
           # PMC bytes per launch at 16 clips x 16 frames (tools/pmc_traffic.sh), scaled to this launch's positions
           "traffic": None if pmc_bytes_16x16 is None else pmc_bytes_16x16 * (iso["clips"] * T) / (16.0 * 16.0),
           "algorithmic_bytes_per_launch": iso["clips"] * T * 56 * 56 * byte_pos}
    if live is not None:
        out["in_step_live"] = live if "error" in live else dict(
            rates(live["ms"], live["clips"]), launches=live["launches"], min_ms=live["min_ms"], max_ms=live["max_ms"], how=live["how"])
    if frozen is not None:
        out["in_step"] = dict(frozen, **{k: v for k, v in rates(frozen["ms"], frozen["clips"]).items() if k != "ms"})
    return out


def bf16_leg(a, rank, world, local, dev):
    """BASELINE configs[4] ("cfg5"): large-batch stress on the 16-bit MFMA path -- per-GPU batch 128 x 32-frame clips
    (global 1024 on 8 GPUs), both trunks in bf16 (fp32 master weights, fp32 BatchNorm statistics), heads fp32 (MFMA)."""
    import torch.distributed as dist
    from selavi_amd import model as smodel, optim, train
    B, T, hc, K = a.cfg5_batch, CFG5["T"], CFG5["hc"], CFG5["K"]
    torch.manual_seed(31)
    m = smodel.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=K,
                          pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).to(dev)
    m.set_precision("bf16")
    m.train()
    net = train.data_parallel(m, [local]) if world > 1 else m
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    video = torch.randn(B, 3, T, CFG5["S"], CFG5["S"], device=dev, generator=g)
    audio = torch.randn(B, 1, CFG5["F"], CFG5["Tp"], device=dev, generator=g)
    selflabels = torch.randint(0, K, (4096, hc), device=dev, generator=g)
    selected = torch.randint(0, 4096, (B,), device=dev, generator=g)
    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(max(a.cfg5_warmup, 1)):
        loss = train.train_step(net, opt, video, audio, selflabels, selected, hc)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    steps = a.cfg5_steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # per-step times without host syncs
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = train.train_step(net, opt, video, audio, selflabels, selected, hc)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    peak = torch.cuda.max_memory_allocated(dev)
    ms = dt / steps * 1e3
    raw_steps = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    print("cfg5 leg per-step ms: " + " ".join("%.1f" % v for v in raw_steps), file=sys.stderr)
    per_step = sorted(raw_steps)
    ms_median, ms_min = per_step[len(per_step) // 2], per_step[0]
    with torch.no_grad():
        for _ in range(2):
            m(video, audio)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(5):
            m(video, audio)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / 5 * 1e3
    from selavi_amd import ops16 as _o16
    # the two layer-1 forward kernels: alone on the chip, and LIVE inside the step (HIP events around each of their launches on
    # the stream they run on, over 2 extra untimed steps -- in the step they share the chip with the audio trunk's stream)
    plans, clips_per_launch = {}, {}
    for name, L in HOT16.items():
        hp = _o16.Plan16.get(B, T, 56, 56, L["cin"], L["cout"], L["k"], (1, 1, 1), L["pad"], dev)
        if hp.chunks is None:
            plans[name], clips_per_launch[name] = hp, B
        else:                               # sliced at the 32-bit buffer range: the first slice's launches
            plans[name], clips_per_launch[name] = hp.chunks[0][2], hp.chunks[0][1] - hp.chunks[0][0]
    live = {}
    try:
        with _o16.probe_conv_fwd(plans, prologue=True) as pr:
            for _ in range(2):
                train.train_step(net, opt, video, audio, selflabels, selected, hc)
        for name, tms in pr.ms().items():
            if tms:
                live[name] = dict(ms=sum(tms) / len(tms), launches=len(tms), min_ms=min(tms), max_ms=max(tms),
                                  clips=clips_per_launch[name],
                                  how="HIP events around the launch on its stream, 2 extra steps after the timed region")
    except Exception as e:          # measurement garnish: never take the leg down
        if world > 1:
            raise
        live = {name: dict(error=repr(e)) for name in HOT16}
    roofs = {}
    for name, L in HOT16.items():
        iso = hot_conv16_roofline(name, min(B, 64), T, dev)
        fz = _rocprof_in_step(L["kernel"], L["grid"], ("r06_step16_cfg5_kernel_summary.txt", "r05_step16_cfg5_kernel_summary.txt")) \
            if B == CFG5["batch"] else None
        if fz is not None:
            fz = dict(fz, clips=clips_per_launch[name])
        roofs[name] = _roof16(name, T, iso, live.get(name), fz, _pmc_traffic(L["pmc"]))
    step_tf = 3 * FWD_GFLOP_PER_CLIP_T32 * B / ms
    step_gbs = 3 * FWD_MB_PER_CLIP_T32_BF16 * B / ms
    loss_v = float(loss.item())
    del m, net, opt
    return {
        "metric": "clips/sec (video+audio fwd/bwd + loss + SGD), 16-bit MFMA path", "value": world * B * steps / dt,
        "unit": "clips/s", "n_gpus": world, "steps": steps, "warmup": max(a.cfg5_warmup, 1), "ms_per_step": ms,
        # per-step durations between HIP events on the step's stream (rank 0's): the leg's spread
        "ms_per_step_median": ms_median, "ms_per_step_min": ms_min, "ms_per_step_max": per_step[-1],
        # (value = the wall clock over all timed steps, as for the headline; one stalled step -- seen once in a dozen runs on
        #  the pool's boxes: 7.8 s -- drags it down, the median does not)
        "value_from_median_step": world * B / ms_median * 1e3, "steps_over_2x_median": sum(v > 2 * ms_median for v in per_step),
        "dtype": "bf16",
        "config": {"workload": "cfg5: R(2+1)D-18 + ResNet-9 in bf16 (fp32 master weights, fp32 BN statistics), heads fp32, "
                               "per-GPU bs=%d, 32x112x112 video, 1x129x100 log-mel, K=309, headcount=10" % B,
                   "global_batch": world * B, "parallelism": "dp%d" % world, "loss_last_step": loss_v,
                   "peak_hbm_gb": round(peak / 2 ** 30, 2)},
        # `roofline` = the HBM-bound one of the two dominant forward kernels (the north star quotes the forward against the HBM
        # roofline); `roofline_mfma_kernel` = the MFMA-bound one (the largest single kernel of the forward by time)
        "roofline": roofs["l1_temporal"],
        "roofline_mfma_kernel": roofs["l1_spatial"],
        "step_roofline": {"mfma": {"achieved": step_tf, "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s per GPU",
                                   "frac": step_tf / PEAK_BF16_MFMA_TF},
                          "hbm": {"achieved": step_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s per GPU (algorithmic)",
                                  "frac": step_gbs / PEAK_HBM_GBS}},
        "forward": {"ms": fwd_ms, "clips_per_s_per_gpu": B / fwd_ms * 1e3,
                    "hbm_frac": FWD_MB_PER_CLIP_T32_BF16 * B / fwd_ms / PEAK_HBM_GBS,
                    "mfma_frac": FWD_GFLOP_PER_CLIP_T32 * B / fwd_ms / PEAK_BF16_MFMA_TF},
    }


def cfg2_bf16_leg(a, dev):
    """The HEADLINE shape (cfg2: 16 clips x 16 frames per GPU) on the 16-bit MFMA path -- what configs[1] costs when the
    convs run in bf16 (fp32 master weights, fp32 BatchNorm statistics; not the parity path).  At this size the step is
    ~700 short kernels: eager it is bound by the host's enqueue time, so the leg also replays it as ONE HIP graph
    (train.GraphedStep: the three streams of the step stay concurrent inside the capture).  Single GPU only."""
    from selavi_amd import model as smodel, optim, train
    B, T, hc, K = CFG2["batch"], CFG2["T"], CFG2["hc"], CFG2["K"]
    torch.manual_seed(31)
    m = smodel.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=K,
                          pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).to(dev)
    m.set_precision("bf16")
    m.train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(99)
    video = torch.randn(B, 3, T, CFG2["S"], CFG2["S"], device=dev, generator=g)
    audio = torch.randn(B, 1, CFG2["F"], CFG2["Tp"], device=dev, generator=g)
    selflabels = torch.randint(0, K, (4096, hc), device=dev, generator=g)
    selected = torch.randint(0, 4096, (B,), device=dev, generator=g)

    def timed(fn, n):
        for _ in range(3):
            loss = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)
    n = max(a.steps, 10)
    eager_ms, loss_e = timed(lambda: train.train_step(m, opt, video, audio, selflabels, selected, hc), n)
    t0 = time.perf_counter()
    train.train_step(m, opt, video, audio, selflabels, selected, hc)
    host_ms = (time.perf_counter() - t0) * 1e3            # enqueue time of one eager step (no device wait)
    torch.cuda.synchronize()
    out = {"metric": "clips/sec (video+audio fwd/bwd + loss + SGD), 16-bit MFMA path at the headline shape", "unit": "clips/s",
           "dtype": "bf16", "steps": n,
           "config": {"workload": "cfg2 shape (bs=16, 16x112x112, 1x129x100, K=309, headcount=10), both trunks bf16, heads fp32",
                      "loss_last_step": loss_e},
           "eager": {"value": B / eager_ms * 1e3, "ms_per_step": eager_ms, "host_enqueue_ms_per_step": host_ms}}
    try:
        gs = train.GraphedStep(m, opt, video, audio, selflabels, selected, hc)
        graph_ms, loss_g = timed(gs.replay, n)
        out["graph_replay"] = {"value": B / graph_ms * 1e3, "ms_per_step": graph_ms, "loss_last_step": loss_g}
    except Exception as e:
        out["graph_replay"] = {"error": repr(e)}
    best = min(eager_ms, out["graph_replay"].get("ms_per_step", eager_ms))
    out["value"], out["ms_per_step"] = B / best * 1e3, best
    out["step_roofline"] = {"hbm": {"frac": 3 * (FWD_MB_PER_CLIP / 2) * B / best / PEAK_HBM_GBS},
                            "mfma": {"frac": 3 * FWD_GFLOP_PER_CLIP * B / best / PEAK_BF16_MFMA_TF}}
    return out


def cpu_baseline(batch, warm=3, timed=3):
    """The oracle (oracle/step_ref.py: torch CPU restatement of main.py:284-302, validated against the executed
    reference) timed on this host's cores at the SAME shapes as the GPU step (cfg2, batch 16),
    `warm` warm-up steps, then `timed` steps: best and median.  ~7 s per step on the GPU box's host (~45 s in all)."""
    from oracle import model_ref, step_ref
    # torch/oneDNN conv3d backward degrades badly when oversubscribed across sockets (256 threads:
    # 210 s/step at batch 2 on the GPU box); 32 threads is the fastest setting measured there.
    cores = min(os.cpu_count(), 32)
    torch.set_num_threads(cores)
    m = model_ref.load_model(use_mlp=True, num_classes=CFG2["K"], norm_feat=False, headcount=CFG2["hc"])
    m.train()
    opt = step_ref.make_optimizer(m)
    g = torch.Generator().manual_seed(0)
    video = torch.randn(batch, 3, CFG2["T"], CFG2["S"], CFG2["S"], generator=g)
    audio = torch.randn(batch, 1, CFG2["F"], CFG2["Tp"], generator=g)
    sl = torch.randint(0, CFG2["K"], (1024, CFG2["hc"]), generator=g)
    sel = torch.randint(0, 1024, (batch,), generator=g)
    t_all = time.time()
    for _ in range(warm):
        step_ref.train_step(m, opt, video, audio, sl, sel, CFG2["hc"])
        if time.time() - t_all > 90.0:        # a slow host: keep the bench line bounded
            break
    ts = []
    for _ in range(timed):
        t0 = time.time()
        step_ref.train_step(m, opt, video, audio, sl, sel, CFG2["hc"])
        ts.append(time.time() - t0)
        if time.time() - t_all > 150.0 and len(ts) >= 1:
            break
    ts.sort()
    best, med = ts[0], ts[len(ts) // 2]
    return dict(value=batch / med, unit="clips/s", cores=cores, kind="port", best=batch / best, median=batch / med,
                sample=f"cfg2 full step (fwd+loss+bwd+SGD) at batch {batch} (the GPU step's shapes), {warm} warm-ups + "
                       f"{len(ts)} timed steps, torch {torch.__version__} CPU fp32 on {cores} threads, "
                       f"best {best:.2f} / median {med:.2f} s/step")


def sk_cpu_baseline(iters=40):
    """The SK half of BASELINE's metric on the host: the oracle's loop (oracle/sk_ref.py <- src/sk_utils.py:400-406, numpy
    fp64, two matrix-vector products over P per iteration) at the VGG-Sound size, `iters` iterations of the loop itself
    (the oracle reports the loop's wall time; the power / label steps around it are not counted)."""
    import numpy as np
    from oracle import sk_ref
    N, K = CFG2["N"], CFG2["K"]
    rng = np.random.default_rng(5)
    PS = rng.random((N, K)) * 0.5 + 0.5           # the loop's cost does not depend on the values (dense fp64 gemv)
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    sk_ref.optimize_L_sk(PS, max_iter=2, tol=-1.0)                   # warm-up (allocator, page cache, BLAS threads)
    info = sk_ref.optimize_L_sk(PS, max_iter=iters, tol=-1.0)[2]     # tol < 0: exactly `iters` iterations
    per = info["loop_s"] / info["iters"]
    return dict(value=1.0 / per, unit="SK iterations/s", cores=cores, kind="port",
                sample=f"oracle/sk_ref.optimize_L_sk loop, N={N}, K={K}, {iters} iterations (numpy {np.__version__} fp64 "
                       f"gemv, {per * 1e3:.0f} ms/iteration)")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # Dry-run knobs for exercising the N > 1 path on a one-GPU box (all ranks on cuda:0 over gloo); the
    # measured configuration is always one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("SELAVI_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("SELAVI_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from selavi_amd import model as smodel, ops, optim, train
    ops.set_benchmark(True)          # main.py:187 cudnn.benchmark = True: per-shape launch-config timing (in warm-up)
    B, hc, K = a.batch, CFG2["hc"], CFG2["K"]
    torch.manual_seed(31)            # opt.py:152
    m = smodel.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=K,
                          pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).to(dev)
    m.train()
    net = m
    if world > 1:
        net = train.data_parallel(m, [local])     # main.py:156-160; SELAVI_DP=ddp selects torch DDP instead
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)       # main.py:132-137
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    video = torch.randn(B, 3, CFG2["T"], CFG2["S"], CFG2["S"], device=dev, generator=g)
    audio = torch.randn(B, 1, CFG2["F"], CFG2["Tp"], device=dev, generator=g)
    selflabels = torch.randint(0, K, (CFG2["N"], hc), device=dev, generator=g)
    selected = torch.randint(0, CFG2["N"], (B,), device=dev, generator=g)

    def step():
        return train.train_step(net, opt, video, audio, selflabels, selected, hc)

    loss = step()        # plan-building pass (benchmark-mode launch-configuration timing), never timed
    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_v = float(loss.item())
    peak_hbm = torch.cuda.max_memory_allocated(dev)      # of the training step; the measurements below allocate more
    ms_step = dt / a.steps * 1e3
    clips = world * B * a.steps / dt

    # forward only (train-mode batch statistics, no autograd state): the north star quotes the R(2+1)D
    # forward against the HBM roofline; at fp32 the forward is MFMA-bound, both are reported
    with torch.no_grad():
        for _ in range(2):
            m(video, audio)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(5):
            m(video, audio)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / 5 * 1e3
    hot = hot_conv_roofline(B, dev)
    # the same launch INSIDE the step, live: HIP events around every forward launch of that layer shape with the BatchNorm +
    # ReLU prologue, on the stream it runs on, over extra (untimed) steps -- in the step it shares the chip with the audio
    # trunk's stream, isolated it runs alone
    in_step_live = None
    try:
        hot_plan = ops.ConvPlan.get((B, 64, 16, 56, 56), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), dev)
        with ops.probe_conv_fwd(hot_plan, prologue=True) as pr:
            for _ in range(3):
                step()
        tms = pr.ms()
        if tms:
            in_step_live = dict(ms=sum(tms) / len(tms), launches=len(tms), min_ms=min(tms), max_ms=max(tms),
                                how="HIP events around the launch on its stream, 3 extra steps after the timed region")
    except Exception as e:          # measurement garnish: never take the line down
        if world > 1:
            raise                   # (but ranks must not diverge inside the step's collectives)
        in_step_live = dict(error=repr(e))
    # N > 1: what the data-parallel exchanges cost, so that the first real multi-GPU line explains itself (main.py:117-118,
    # 156-160; utils.py:133-146): transport and the ranks RCCL reports per communicator, SyncBN exchanges per step and their
    # mean duration, the gradient buckets' bytes and the EXPOSED wait for them -- over 2 extra (untimed) steps
    comm_diag = None
    if world > 1:
        from selavi_amd import comm as scomm
        with scomm.diagnostics() as dg:
            for _ in range(2):
                step()
        comm_diag = dg.report(steps=2)
    sk = None if a.no_sk else sk_bench(rank, world, dev)
    if comm_diag is not None:
        comm_diag = dict(scomm.describe(), wrapper=type(net).__name__, **comm_diag)      # (after SK: its communicator exists now)
        if sk is not None:
            comm_diag["sk"] = dict(us_per_iter=sk["us_per_iter"], **(sk["us_per_iter_split"] or {}))
    sk_round = sk_round_estimate(m, dev, world, clips, sk) if sk else None
    # the same step on the NATIVE fp32-input MFMA kernels (csrc/igemm.hpp, v_mfma_f32_16x16x4_f32): what the headline was in
    # rounds 1-3, for comparison with the split-operand arithmetic the headline runs on now (csrc/igemm3.hpp)
    native = None
    if not a.no_native_leg and ops.conv_arithmetic() == "x3":
        ops.set_conv_arithmetic("native")
        try:
            step()
            for _ in range(2):
                loss_n = step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tn0 = time.perf_counter()
            nsteps = max(3, a.steps // 2)
            for _ in range(nsteps):
                loss_n = step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dtn = time.perf_counter() - tn0
            if world > 1:
                t = torch.tensor([dtn], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtn = t.item()
            hot_n = hot_conv_roofline(B, dev)
            native = {"value": world * B * nsteps / dtn, "unit": "clips/s", "steps": nsteps, "ms_per_step": dtn / nsteps * 1e3,
                      "loss_last_step": float(loss_n.item()),
                      "hot_kernel": {"kernel": hot_n["kernel"], "ms_per_launch": hot_n["ms"], "achieved": hot_n["tflops"],
                                     "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s", "frac": hot_n["tflops"] / PEAK_FP32_MFMA_TF},
                      "note": "SELAVI_CONV_X3=0 / ops.set_conv_arithmetic('native'): every conv on v_mfma_f32_16x16x4_f32"}
        finally:
            ops.set_conv_arithmetic("x3")
    cfg5 = None
    if not a.no_cfg5:
        del video, audio
        m.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()
        try:
            cfg5 = bf16_leg(a, rank, world, local, dev)
        except Exception as e:                       # a second leg: never take the headline line down with it
            if world > 1:
                raise                                # (but ranks must not diverge inside collectives)
            cfg5 = {"error": repr(e)}
    cfg2_16 = None
    if world == 1 and not a.no_cfg5:
        try:
            cfg2_16 = cfg2_bf16_leg(a, dev)
        except Exception as e:                       # an extra leg: never take the headline line down with it
            cfg2_16 = {"error": repr(e)}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_batch)
        if sk is not None:
            sk["cpu_baseline"] = sk_cpu_baseline()

    if rank == 0:
        step_tflops = 3 * FWD_GFLOP_PER_CLIP * B / ms_step          # per GPU, algorithmic
        out = {
            "metric": "clips/sec (video+audio fwd/bwd + loss + SGD; SK timed separately)",
            "value": clips, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # BASELINE's metric names the SK rounds too: whole-run clips/s with the pseudo-label rounds of the reference's
            # schedule folded in (sk_round below has the derivation; bf16 = the opt-in 16-bit feature pass)
            "clips_per_s_including_sk": None if not sk_round else sk_round["clips_per_s_including_sk"],
            "clips_per_s_including_sk_bf16_feature_pass": None if not (sk_round and sk_round.get("bf16_feature_pass_opt_in"))
            else sk_round["bf16_feature_pass_opt_in"]["clips_per_s_including_sk"],
            "clips_per_s_including_sk_fp32x2_feature_pass": None if not (sk_round and "fp32x2" in sk_round.get("by_feature_pass", {}))
            else sk_round["by_feature_pass"]["fp32x2"]["clips_per_s_including_sk"],
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: R(2+1)D-18 + ResNet-9, per-GPU bs=%d, 16x112x112 video, 1x129x100 "
                                   "log-mel, K=309, headcount=10, SGD(m=0.9, wd=1e-5), fp32" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "conv_arithmetic": ("fp32 tensors and fp32 accumulation; every conv product of all but the two 3 / 1-channel "
                                           "stem launches is evaluated on the bf16 matrix cores from operands cut EXACTLY into three "
                                           "bf16 pieces (a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1: the dropped terms are <= "
                                           "3 x 2^-24 |a b|, one fp32 rounding; per-op error against fp64 equal to the native fp32 "
                                           "MFMA kernels': tests/test_ops_gpu.py) -- csrc/igemm3.hpp; native_fp32_mfma = the same "
                                           "step on v_mfma_f32_16x16x4_f32") if hot["x3"] else "native fp32-input MFMA",
                       "launch_configs": ("benchmark mode: every (tile, K-split) candidate timed once per layer shape in the "
                                          "untimed plan-building step -- main.py:187 cudnn.benchmark = True; tests, smoke() and "
                                          "library users without ops.set_benchmark(True) run the built-in heuristics (~3-5 % slower)"),
                       "library_digest": (_library_digest() or "")[:16],
                       "sync_bn": world > 1, "loss_last_step": loss_v,
                       "peak_hbm_gb": round(peak_hbm / 2 ** 30, 2)},
            "roofline": {"bound": "mfma", "achieved": hot["tflops"],
                         # x3: an fp32 conv FLOP costs six bf16 MFMA FLOPs -> peak = dense bf16 MFMA peak / 6
                         "peak": PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF, "unit": "TFLOP/s (fp32 conv FLOPs)",
                         "frac": hot["tflops"] / (PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF),
                         "peak_note": ("2500 TFLOP/s dense bf16 MFMA / 6 partial products per fp32 product; the native fp32-input "
                                       "MFMA peak is 157.3") if hot["x3"] else "fp32-input MFMA = fp32 vector peak",
                         "frac_of_native_fp32_mfma_peak": hot["tflops"] / PEAK_FP32_MFMA_TF,
                         "bf16_mfma_issued": {"achieved": 6 * hot["tflops"], "peak": PEAK_BF16_MFMA_TF_DENSE,
                                              "frac": 6 * hot["tflops"] / PEAK_BF16_MFMA_TF_DENSE} if hot["x3"] else None,
                         # HBM bytes per launch of this kernel at B=16 from rocprofv3 PMC passes
                         # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), profiles/r04_pmc.json
                         "traffic": _pmc_traffic("hot_conv_fwd_x3" if hot["x3"] else "hot_conv_fwd") if B == CFG2["batch"] else None,
                         "kernel": hot["kernel"], "ms_per_launch": hot["ms"], "flop_per_launch": hot["flop"],
                         "in_step_live": (lambda r: r if (r is None or "error" in r) else dict(
                             r, achieved=hot["flop"] / r["ms"] / 1e9,
                             frac=hot["flop"] / r["ms"] / 1e9 / (PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF)))(in_step_live),
                         # the same launch inside the step (rocprofv3 average of the committed trace of this command; null
                         # unless that trace was taken on the library build this run loaded)
                         "in_step": (lambda r: None if r is None else dict(
                             r, achieved=hot["flop"] / r["ms"] / 1e9,
                             frac=hot["flop"] / r["ms"] / 1e9 / (PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF)))(
                             _rocprof_in_step(HOT_X3_KERNEL if hot["x3"] else "igemm_kernel<0, 9, 2, true, 1, 1, 0, 0, 0>", 6272,
                                              ("r06_bench_kernel_summary.txt", "r05_bench_kernel_summary.txt"))
                             if B == CFG2["batch"] else None)},
            "step_roofline": {"bound": "mfma", "achieved": step_tflops, "peak": PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF,
                              "unit": "TFLOP/s per GPU (algorithmic 3 x %.2f GFLOP/clip)" % FWD_GFLOP_PER_CLIP,
                              "frac": step_tflops / (PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF),
                              "frac_of_native_fp32_mfma_peak": step_tflops / PEAK_FP32_MFMA_TF},
            "forward": {"ms": fwd_ms, "clips_per_s_per_gpu": B / fwd_ms * 1e3,
                        # (priced against the peak the headline runs on: dense bf16 MFMA / 6 for the split-operand convs)
                        "mfma": {"achieved": FWD_GFLOP_PER_CLIP * B / fwd_ms, "peak": PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF,
                                 "unit": "TFLOP/s", "frac": FWD_GFLOP_PER_CLIP * B / fwd_ms / (PEAK_X3_TF if hot["x3"] else PEAK_FP32_MFMA_TF),
                                 "frac_of_native_fp32_mfma_peak": FWD_GFLOP_PER_CLIP * B / fwd_ms / PEAK_FP32_MFMA_TF},
                        "hbm": {"achieved": FWD_MB_PER_CLIP * B / fwd_ms, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": FWD_MB_PER_CLIP * B / fwd_ms / PEAK_HBM_GBS,
                                "note": "algorithmic fused-forward bytes (SURVEY 8d: 518.1 + 3.4 MB/clip); the fp32 "
                                        "forward is MFMA-bound, its compute ceiling is 12.5 % of the HBM roofline"}},
            "comm": comm_diag,
            "sk": sk,
            "sk_round": sk_round,
            "cfg5_bf16": cfg5,
            "cfg2_bf16": cfg2_16,
            "native_fp32_mfma": native,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
