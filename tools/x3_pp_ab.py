"""Split-operand conv kernels (csrc/igemm3.hpp): the 4-wave tiles (nt = 1, 2) against the 8-wave PING-PONG tiles (nt = 4), per
layer shape, forward (BatchNorm + ReLU prologue) and backward data, interleaved rounds inside one process.
Usage: python tools/x3_pp_ab.py [layer-substring] [rounds=3] [batch=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import ops
from tools.conv_bench_layers import LAYERS

sel = sys.argv[1] if len(sys.argv) > 1 else ""
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def name_of(cfg):
    return "heur" if cfg == 0 else f"{(cfg & 255) * 16}x{((cfg >> 8) & 15) * 64}" + (f"/k{cfg >> 16}" if (cfg >> 16) > 1 else "")


for name, Cin, T, H, W, Cout, k, st, pd in LAYERS:
    if sel not in name or name == "stem.0":
        continue
    plan = ops.ConvPlan.get((B, Cin, T, H, W), Cout, k, st, pd, dev)
    x = torch.randn(B, Cin, T, H, W, device=dev, generator=g)
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) * 0.05
    ss = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1])
    y, _, _ = ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True)
    dy = torch.randn_like(y)
    wf, wt = ops.conv_w_transform(plan, w)
    flop = 2.0 * y.numel() * Cin * k[0] * k[1] * k[2]
    out = []
    for op, label in ((0, "fwd"), (1, "dgrad")):
        cands = [0] + [c for c in plan.candidates(op) if (c >> 16) == 1]          # unsplit candidates
        best = {}
        for r in range(rounds):
            for c in cands:
                cfgs = [0, 0, 0]
                cfgs[op] = c
                try:
                    plan.set_configs(*cfgs)
                except ValueError:
                    continue
                t = timeit((lambda: ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)) if op == 0 else
                           (lambda: ops.conv_dgrad(plan, dy, wt)))
                best[c] = min(best.get(c, 1e9), t)
        plan.set_configs(0, 0, 0)
        four = {c: t for c, t in best.items() if c and ((c >> 8) & 15) < 4}
        eight = {c: t for c, t in best.items() if c and ((c >> 8) & 15) == 4}
        b4 = min(four.items(), key=lambda kv: kv[1]) if four else (0, float("nan"))
        b8 = min(eight.items(), key=lambda kv: kv[1]) if eight else (0, float("nan"))
        out.append(f"{label}: heur {best[0]:.3f}  4-wave best {name_of(b4[0])} {b4[1]:.3f} ({flop/b4[1]/1e9:.0f} TF)  "
                   f"8-wave ping-pong best {name_of(b8[0])} {b8[1]:.3f} ({flop/b8[1]/1e9:.0f} TF)")
    print(f"{name:14s} {flop/1e9:7.1f} GF | " + " | ".join(out), flush=True)
