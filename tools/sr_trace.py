"""Where a tile's time goes in conv_cl16_sr_kernel (library built with -DSLV_SR_TRACE: tools/build_variant.sh sr_trace
conv_cl16_sr.hip -- -DSLV_SR_TRACE): s_memtime ticks (10 ns) per section, summed over a workgroup's tiles, per wave.  The
traced build overwrites the head of its output.  Usage: python tools/sr_trace.py [clips] [pro 0/1] [stats 0/1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops16

class Conv:
    in_channels, out_channels, kernel3, stride3, padding3 = 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pro = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stats = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = ops16.to_channels_last16(torch.randn(B, 64, 16, 56, 56, device=dev, generator=g))
plan = ops16.plan_for(x, Conv)
w = torch.randn(144, 64, 1, 3, 3, device=dev, generator=g) * 0.05
ss = torch.stack([torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1]).contiguous()
wf, _ = ops16.conv_w_transform(plan, w)
for _ in range(3):
    y = ops16.conv_fwd(plan, x, w, in_ss=ss if pro else None, in_relu=bool(pro), want_stats=bool(stats), wf=wf)[0]
torch.cuda.synchronize()
tr = y.view(torch.uint8).flatten()[:256 * 4 * 8 * 8].view(torch.int64).cpu().numpy().reshape(256, 4, 8).astype(np.float64)
tiles = B * 16 * 49 / 256
print(f"{B} clips, pro {pro}, stats {stats}: {tiles:.0f} tiles per workgroup; ticks (10 ns) PER TILE, mean over the 256 workgroups")
m = tr[:, :3].mean(axis=(0, 1)) / tiles
print(f"  MFMA waves : MFMAs + items {m[0]:7.1f}   barrier wait {m[1]:7.1f}   (tail {tr[:, :3, 2].mean():.0f} ticks once)")
d = tr[:, 3].mean(axis=0) / tiles
print(f"  data wave  : wait for patch {d[0]:6.1f}   BatchNorm + LDS writes {d[1]:6.1f}   output tile -> memory {d[2]:6.1f}   "
      f"patch requests {d[3]:6.1f}   lgkmcnt {d[4]:6.1f}   barrier wait {d[5]:6.1f}   (tail {tr[:, 3, 6].mean():.0f} once)")
print(f"  per-workgroup total (MFMA waves): min {tr[:, :3, :2].sum(axis=2).min():.0f}  max {tr[:, :3, :2].sum(axis=2).max():.0f} ticks")
