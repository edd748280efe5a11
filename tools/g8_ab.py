"""A/B of the 8-wave kernel of the wide layers (csrc/conv_cl16_g8.hip) against the kernels it replaces (patch kernel
csrc/conv_cl16_s3.hip / tile kernel csrc/conv_cl16.hip), per layer shape, INSIDE one process (interleaved rounds, same
tensors): train-mode forward (BatchNorm + ReLU prologue + statistics), plain forward, backward data.
Usage: python tools/g8_ab.py [batch=64] [frames=16] [rounds=3] [layer-substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import ops16
from selavi_amd._lib import C
from tools.conv_bench_layers import LAYERS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T0 = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sel = sys.argv[4] if len(sys.argv) > 4 else ""
NEW = int(os.environ.get("G8_NEW_MODE", "1"))      # mode of the "new" column: 1 = the dispatch rule, 2 = the kernel on every launch it can express
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"batch {B}, {T0} frames; ms = best of {rounds} interleaved rounds x 5 launches   (old = g8 off, new = g8 mode 1, s3-eligible launches included)")
print(f"{'layer':14s} {'GFLOP':>8s} | {'fwd old':>8s} {'new':>8s} {'TF old':>7s} {'new':>6s} | {'plain old':>9s} {'new':>8s} | {'dgrad old':>9s} {'new':>8s} {'TF old':>7s} {'new':>6s}")
tot = {k: 0.0 for k in ("fo", "fn", "do", "dn")}
for name, Cin, T, H, W, Cout, k, st, pd in LAYERS:
    if sel not in name or name.startswith("stem") or name.startswith("l1"):
        continue
    T = T * T0 // 16
    x = ops16.to_channels_last16(torch.randn(B, Cin, T, H, W, device=dev, generator=g))
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) * 0.05
    ss = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1]).contiguous()
    res = {}
    plans = {}
    for mode in (0, NEW):
        C.slv_cl16_g8_mode(mode)
        ops16.Plan16._cache.clear()
        plan = ops16.plan_for(x, Conv(Cin, Cout, k, st, pd))
        wf, wt = ops16.conv_w_transform(plan, w)
        y, _, _ = ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
        plans[mode] = (plan, wf, wt, y)
    dy = torch.randn(plans[0][3].shape, device=dev, generator=g).to(torch.bfloat16)
    dy[..., Cout:] = 0
    # same results (the summation order differs: a few last-place bf16 flips)
    d = (plans[0][3].float() - plans[NEW][3].float()).abs().max().item() / (plans[0][3].float().abs().max().item() + 1e-30)
    for r in range(rounds):
        for mode in (0, NEW):
            C.slv_cl16_g8_mode(mode)
            plan, wf, wt, _ = plans[mode]
            t = (timeit(lambda: ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)),
                 timeit(lambda: ops16.conv_fwd(plan, x, w, want_stats=False, wf=wf)),
                 timeit(lambda: ops16.conv_dgrad(plan, dy, wt)))
            res[mode] = t if mode not in res else tuple(min(a, b) for a, b in zip(res[mode], t))
    flop = 2.0 * B * plans[0][0].out_dims[0] * plans[0][0].out_dims[1] * plans[0][0].out_dims[2] * Cout * Cin * k[0] * k[1] * k[2]
    o, n = res[0], res[NEW]
    print(f"{name:14s} {flop/1e9:8.1f} | {o[0]:8.3f} {n[0]:8.3f} {flop/o[0]/1e9:7.0f} {flop/n[0]/1e9:6.0f} | {o[1]:9.3f} {n[1]:8.3f} | "
          f"{o[2]:9.3f} {n[2]:8.3f} {flop/o[2]/1e9:7.0f} {flop/n[2]/1e9:6.0f}   max rel diff {d:.1e}")
    tot["fo"] += o[0]; tot["fn"] += n[0]; tot["do"] += o[2]; tot["dn"] += n[2]
print(f"sum (each layer once): fwd {tot['fo']:.3f} -> {tot['fn']:.3f} ms, dgrad {tot['do']:.3f} -> {tot['dn']:.3f} ms")
C.slv_cl16_g8_mode(1)
