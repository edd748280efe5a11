"""Where a step's time goes in conv_cl16_tr_kernel (library built with -DSLV_TR_TRACE: tools/build_variant.sh tr_trace
conv_cl16_tr.hip -- -DSLV_TR_TRACE): s_memtime ticks (= shader cycles) per section, per (32-pixel, one-frame) step, mean
over the waves.  The traced build overwrites the head of its output.  Usage: python tools/tr_trace.py [fwd|dgrad] [clips]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops16

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
names = ["requests issued", "MFMAs (+ fragment reads)", "staged frame: wait + BatchNorm/ReLU", "staged frame -> LDS",
         "statistics", "output rows -> memory", None, None, "settle nops", "rounding + output tile -> LDS"]
if mode == "fwd":
    class Conv:
        in_channels, out_channels, kernel3, stride3, padding3 = 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)
    x = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
    plan = ops16.plan_for(x, Conv)
    w = torch.randn(64, 144, 3, 1, 1, device=dev, generator=g) * 0.05
    ss = torch.stack([torch.rand(144, device=dev, generator=g) + 0.5, torch.randn(144, device=dev, generator=g) * 0.1]).contiguous()
    wf, _ = ops16.conv_w_transform(plan, w)
    for _ in range(3):
        y = ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, want_stats=True, wf=wf)[0]
elif mode == "fwd_plain":
    class Conv:
        in_channels, out_channels, kernel3, stride3, padding3 = 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)
    x = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
    plan = ops16.plan_for(x, Conv)
    w = torch.randn(64, 144, 3, 1, 1, device=dev, generator=g) * 0.05
    wf, _ = ops16.conv_w_transform(plan, w)
    for _ in range(3):
        y = ops16.conv_fwd(plan, x, w, want_stats=False, wf=wf)[0]
else:
    class Conv:
        in_channels, out_channels, kernel3, stride3, padding3 = 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)
    x = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
    plan = ops16.plan_for(x, Conv)
    w = torch.randn(64, 144, 3, 1, 1, device=dev, generator=g) * 0.05
    _, wt = ops16.conv_w_transform(plan, w)
    dy = ops16.to_channels_last16(torch.randn(B, 64, 16, 56, 56, device=dev, generator=g))
    b5 = torch.randn(5, 144, device=dev, generator=g) * 0.1
    for _ in range(3):
        y = ops16.conv_dgrad(plan, dy, wt, bn_apply=(x, b5)) if mode == "dgrad" else ops16.conv_dgrad(plan, dy, wt)
torch.cuda.synchronize()
tr = y.view(torch.uint8).flatten()[:1024 * 10 * 8].view(torch.int64).cpu().numpy().reshape(1024, 10).astype(np.float64)
steps = tr[:, 7]
print(f"{mode}, {B} clips: {steps.mean():.0f} steps per wave; cycles PER STEP, mean over 1024 waves")
tot = 0
for i, n in enumerate(names):
    if n is None:
        continue
    v = (tr[:, i] / steps).mean(); tot += v
    print(f"  {n:40s} {v:8.1f}")
print(f"  {'sum':40s} {tot:8.1f}")
