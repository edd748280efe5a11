"""Per-stage timeline of conv_cl16_s3_kernel on the layer-1 spatial forward (library built with -DSLV_S3_TRACE:
tools/s3_ablate.sh trace): cycles between the 5 trace points of a stage, averaged over stages 1..7 and 64 tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops16

class Conv:
    in_channels, out_channels, kernel3, stride3, padding3 = 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = ops16.to_channels_last16(torch.randn(16, 64, 16, 56, 56, device=dev, generator=g))
plan = ops16.plan_for(x, Conv)
w = torch.randn(144, 64, 1, 3, 3, device=dev, generator=g) * 0.05
ss = torch.stack([torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1]).contiguous()
wf, _ = ops16.conv_w_transform(plan, w, need_wt=False)
from selavi_amd._lib import C, ptr, stream
y = torch.empty(plan.out_shape, dtype=torch.bfloat16, device=dev)
ssum = torch.empty(plan.Cout * plan.nblk, dtype=torch.float32, device=dev)
ssq = torch.zeros(plan.Cout * plan.nblk + 64 * 128, dtype=torch.float32, device=dev)      # statistics + the trace area
for _ in range(3):
    C.slv_cl16_conv(plan.g_fwd.ctypes.data, plan.mt_f, ptr(x), ptr(wf), ptr(y), ptr(ss), 0, 0, 0, ptr(ssum), ptr(ssq),
                    0, 0, 0, 0, 0, 0, stream())
torch.cuda.synchronize()
raw = ssq[plan.Cout * plan.nblk:].view(torch.int64).cpu().numpy().reshape(64, 64)        # 64 tiles x 64 uint64
tr = raw[:, :45].reshape(64, 9, 5).astype(np.float64)
d = np.diff(tr, axis=2)                                 # issue loads | compute | store_a | barrier
nxt = tr[:, 1:, 0] - tr[:, :-1, 4]
names = ["issue weight loads (LDS-DMA)", "chunk 0: reads + 18 MFMA + refills", "chunk 1: 18 MFMA", "wait loads + barrier"]
print("cycles per stage (mean over stages 1..7 and 64 tiles; s_memtime ticks = shader cycles):")
for i, n in enumerate(names):
    print(f"  {n:36s} {d[:, 1:8, i].mean():8.0f}  (min {d[:, 1:8, i].min():.0f}, max {d[:, 1:8, i].max():.0f})")
print(f"  {'loop back':36s} {nxt[:, 1:7].mean():8.0f}")
print(f"  whole stage                          {(tr[:, 8, 4] - tr[:, 0, 0]).mean() / 9:8.0f}")
k = raw[:, 60:64].astype(np.float64)
print(f"per tile: pipeline prologue (patch + first weights) {(k[:, 1] - k[:, 0]).mean():.0f}, K loop {(k[:, 2] - k[:, 1]).mean():.0f}, "
      f"epilogue (transpose, statistics, stores landed) {(k[:, 3] - k[:, 2]).mean():.0f}")
print("  tile 0, stage 0..8 totals:", (tr[0, :, 4] - tr[0, :, 0]).astype(int).tolist())
