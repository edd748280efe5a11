"""Per-tile timeline of cl16_wgrad_acc_kernel on the layer-1 spatial weight gradient (library built with -DSLV_WA_TRACE:
tools/build_variant.sh watrace wgrad_cl16_acc.hip -- -DSLV_WA_TRACE): cycles between 5 trace points of tiles 8..23, every
wave of blocks 0..15.  The traced build does not write its result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops16, ops as _ops

class Conv:
    in_channels, out_channels, kernel3, stride3, padding3 = 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = ops16.to_channels_last16(torch.randn(B, 64, 16, 56, 56, device=dev, generator=g))
plan = ops16.plan_for(x, Conv)
dy = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
ss = torch.stack([torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1]).contiguous()
for _ in range(3):
    ops16.conv_wgrad(plan, dy, x, in_ss=ss, in_relu=True)
torch.cuda.synchronize()
ws = _ops.workspace(plan.ws_wgrad, dy.device)
tr = ws.view(torch.uint8)[:16 * 4 * 16 * 5 * 8].view(torch.int64).cpu().numpy().reshape(16, 4, 16, 5).astype(np.float64)
d = np.diff(tr, axis=3)
names = ["K step 0 (81 MFMA + requests of tile k+2)", "K step 1 (81 MFMA + staging stores)", "s_waitcnt vmcnt(0) lgkmcnt(0)", "barrier"]
print("cycles (s_memtime ticks) per tile, mean over tiles 8..23 of 16 blocks, per wave:")
for i, n in enumerate(names):
    print(f"  {n:44s} " + "  ".join(f"w{w} {d[:, w, :, i].mean():7.0f}" for w in range(4)) + f"   (min {d[:, :, :, i].min():.0f}, max {d[:, :, :, i].max():.0f})")
print(f"  {'loop back':44s} {(tr[:, :, 1:, 0] - tr[:, :, :-1, 4]).mean():8.0f}")
print(f"  whole tile                                   {(tr[:, :, 15, 4] - tr[:, :, 0, 0]).mean() / 15.8:8.0f}")
print("one block, wave 0, tiles 8..11 (ticks since tile 8 start):"); print((tr[0, 0, :4] - tr[0, 0, 0, 0]).astype(int))
