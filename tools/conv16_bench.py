"""Per-layer timing of the bf16 channels-last conv ops (16-bit MFMA path) at cfg2 shapes (B=16 by default): TFLOP/s of the
train-mode forward (BN+ReLU prologue, statistics epilogue), backward data and weight gradient, plus the HBM-bound
BatchNorm kernels.  Usage: python tools/conv16_bench.py [layer-substring] [reps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import ops16
from tools.conv_bench_layers import LAYERS

sel = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{'layer':14s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s} | apply / reduce / act ms (GB/s)")
tot = [0.0, 0.0, 0.0, 0.0]
for name, Cin, T, H, W, Cout, k, st, pd in LAYERS:
    if sel not in name:
        continue
    stem = name == "stem.0"
    if stem:
        x = torch.randn(B, Cin, T, H, W, device=dev, generator=g)
    else:
        x = ops16.to_channels_last16(torch.randn(B, Cin, T, H, W, device=dev, generator=g))
    plan = ops16.plan_for(x, Conv(Cin, Cout, k, st, pd))
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) * 0.05
    ss = None if stem else torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5,
                                        torch.randn(Cin, device=dev, generator=g) * 0.1]).contiguous()
    wf, wt = ops16.conv_w_transform(plan, w)
    y, _, _ = ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=ss is not None, wf=wf)
    dy = torch.randn(y.shape, device=dev, generator=g).to(torch.bfloat16)
    dy[..., Cout:] = 0
    b5 = torch.randn(5, Cout, device=dev, generator=g) * 0.1
    mi = torch.stack([torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)]).contiguous()
    flop = 2.0 * B * plan.out_dims[0] * plan.out_dims[1] * plan.out_dims[2] * Cout * Cin * k[0] * k[1] * k[2]
    tf = timeit(lambda: ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=ss is not None, wf=wf))
    tf_plain = timeit(lambda: ops16.conv_fwd(plan, x, w, want_stats=False, wf=wf))          # no prologue, no statistics
    tf_stats = timeit(lambda: ops16.conv_fwd(plan, x, w, want_stats=True, wf=wf))           # statistics only
    dxo = torch.empty_like(dy)
    ta = timeit(lambda: ops16.bn_bwd_apply(dy, y, b5, True, out=dxo))
    gam = torch.ones(Cout, device=dev)
    tr = timeit(lambda: ops16.bn_bwd(dy, y, mi, gam, ss_mask=b5[:2].contiguous()))
    tact = timeit(lambda: ops16.bn_act(y, b5[:2].contiguous(), relu=True))
    td = float("nan") if stem else timeit(lambda: ops16.conv_dgrad(plan, dxo, wt))
    tw = timeit(lambda: ops16.conv_wgrad(plan, dxo, x, in_ss=ss, in_relu=ss is not None))
    nbytes = y.numel() * 2
    print(f"{name:14s} {flop/1e9:8.1f} | {tf:8.3f} {flop/tf/1e9:6.1f} | {td:8.3f} {flop/td/1e9:6.1f} | {tw:8.3f} {flop/tw/1e9:6.1f} | "
          f"{ta:6.3f} ({3*nbytes/ta/1e6:5.0f}) {tr:6.3f} ({2*nbytes/tr/1e6:5.0f}) {tact:6.3f} ({2*nbytes/tact/1e6:5.0f})"
          f"  wgrad tile {plan.wm*32}x{plan.wn*32} slices {plan.g_wgrad[22]} | fwd plain {tf_plain:.3f} stats-only {tf_stats:.3f}")
    tot[0] += flop; tot[1] += tf; tot[2] += 0 if stem else td; tot[3] += tw
print(f"sum over listed layers (each once): fwd {tot[1]:.3f} ms, dgrad {tot[2]:.3f} ms, wgrad {tot[3]:.3f} ms")
