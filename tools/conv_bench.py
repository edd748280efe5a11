"""Per-layer timing of the implicit-GEMM conv ops at cfg2 shapes (B=16): TFLOP/s of fwd / dgrad / wgrad
with the fused BN prologues the engine uses.  Usage: python tools/conv_bench.py [layer-substring] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import ops

B = 16
from tools.conv_bench_layers import LAYERS
if os.environ.get("TUNE", "0") == "1":      # time with the benchmark-mode (autotuned) configurations
    ops.benchmark = True
sel = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
print(f"{'layer':14s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
for name, Cin, T, H, W, Cout, k, st, pd in LAYERS:
    if sel not in name:
        continue
    plan = ops.ConvPlan.get((B, Cin, T, H, W), Cout, k, st, pd, dev)
    x = torch.randn(B, Cin, T, H, W, device=dev, generator=g)
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) * 0.05
    ss = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1])
    y, _, _ = ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True)
    dy = torch.randn_like(y)
    b5 = torch.randn(5, Cout, device=dev, generator=g) * 0.1
    wf, wt = ops.conv_w_transform(plan, w)
    flop = 2.0 * y.numel() * Cin * k[0] * k[1] * k[2]
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tf = timeit(lambda: ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf))
    dxo = torch.empty_like(dy)
    ta = timeit(lambda: ops.bn_bwd_apply(dy, y, b5, True, out=dxo))
    td = timeit(lambda: ops.conv_dgrad(plan, dxo, wt))
    tw = timeit(lambda: ops.conv_wgrad(plan, dxo, x, in_ss=ss, in_relu=True))
    if ops.benchmark:
        print("   tuned:", ops._tune_log[-1][2:])
    print(f"{name:14s} {flop/1e9:8.1f} | {tf:8.3f} {flop/tf/1e9:6.1f} | {td:8.3f} {flop/td/1e9:6.1f} | {tw:8.3f} {flop/tw/1e9:6.1f} | apply {ta:6.3f}")
