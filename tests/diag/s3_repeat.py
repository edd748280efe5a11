"""Bit-reproducibility of the bf16 patch conv kernel: the same launch repeated, outputs compared with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from selavi_amd import ops16

class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (N, Cin, T, H, W, Cout) in [(2, 64, 4, 28, 28, 144), (2, 144, 4, 14, 14, 64), (2, 128, 2, 14, 14, 230), (1, 256, 2, 7, 7, 460), (4, 64, 8, 56, 56, 144)]:
    x = ops16.to_channels_last16(torch.randn(N, Cin, T, H, W, device=dev, generator=g))
    plan = ops16.plan_for(x, Conv(Cin, Cout, (1, 3, 3), (1, 1, 1), (0, 1, 1)))
    w = torch.randn(Cout, Cin, 1, 3, 3, device=dev, generator=g) * 0.05
    wf, wt = ops16.conv_w_transform(plan, w)
    ss = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1]).contiguous()
    dy = ops16.to_channels_last16(torch.randn(N, Cout, T, H, W, device=dev, generator=g))
    ref = None
    bad = [0, 0, 0]
    for it in range(30):
        y, s1, s2 = ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, want_stats=True, wf=wf)
        dx = ops16.conv_dgrad(plan, dy, wt)
        cur = (y.clone(), s1.clone(), dx.clone())
        if ref is None:
            ref = cur
        else:
            for i in range(3):
                bad[i] += int(not torch.equal(cur[i], ref[i]))
    print((N, Cin, T, H, W, Cout), "mismatching repeats of 29: fwd", bad[0], "stats", bad[1], "dgrad", bad[2])
