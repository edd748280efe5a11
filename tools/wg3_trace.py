"""Per-step timeline of cl16_wgrad3_kernel on the layer-1 spatial weight gradient (library built with -DSLV_WG3_TRACE:
tools/build_variant.sh wg3trace wgrad_cl16_s3.hip -- -DSLV_WG3_TRACE): cycles between 6 trace points of steps 8..15,
wave 0 of blocks 0..31.  The traced build does not write its result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops16, ops as _ops

CIN, COUT, HW = (int(v) for v in os.environ.get("WG3_SHAPE", "64,144,56").split(","))      # e.g. 128,230,28 = layer 2.1
class Conv:
    in_channels, out_channels, kernel3, stride3, padding3 = CIN, COUT, (1, 3, 3), (1, 1, 1), (0, 1, 1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = ops16.to_channels_last16(torch.randn(B, CIN, 16, HW, HW, device=dev, generator=g))
plan = ops16.plan_for(x, Conv)
dy = ops16.to_channels_last16(torch.randn(B, COUT, 16, HW, HW, device=dev, generator=g))
ss = torch.stack([torch.rand(CIN, device=dev, generator=g) + 0.5, torch.randn(CIN, device=dev, generator=g) * 0.1]).contiguous()
for _ in range(3):
    ops16.conv_wgrad(plan, dy, x, in_ss=ss, in_relu=True)
torch.cuda.synchronize()
ws = _ops.workspace(plan.ws_wgrad, dy.device)
raw = ws.view(torch.uint8)[:32 * 64 * 8].view(torch.int64).cpu().numpy().reshape(32, 64)
tr = raw[:, :48].reshape(32, 8, 6).astype(np.float64)
d = np.diff(tr, axis=2)
names = ["activation-row request (decode + 1 load)", "dY tile request (LDS-DMA x3)", "fragment reads + 30 MFMA",
         "activation rows -> patch (prologue math)", "wait + barrier"]
print("cycles per step (mean over steps 8..15 of 32 blocks; each trace point costs ~100 itself):")
for i, n in enumerate(names):
    print(f"  {n:44s} {d[:, :, i].mean():8.0f}  (min {d[:, :, i].min():.0f}, max {d[:, :, i].max():.0f})")
print(f"  {'loop back':44s} {(tr[:, 1:, 0] - tr[:, :-1, 5]).mean():8.0f}")
print(f"  whole step                                   {(tr[:, 7, 5] - tr[:, 0, 0]).mean() / 8:8.0f}")
if os.environ.get("WG3_TRACE_RAW"):
    print(raw[0, :14]); print(raw[0, 56:64]); print(raw[1, :8]); print(ws.numel() * 4, plan.ws_wgrad)
