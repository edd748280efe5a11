"""Which terms does the bf16 patch kernel's backward data get wrong?  dY nonzero in one 32-channel chunk at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from selavi_amd import ops16

class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd
N, Cin, T, H, W, Cout = 2, 64, 4, 12, 12, 144
k, st, pd = (1, 3, 3), (1, 1, 1), (0, 1, 1)
g = torch.Generator().manual_seed(1)
bf = lambda t: t.to(torch.bfloat16).float()
w = bf(torch.randn(Cout, Cin, *k, generator=g) * 0.03)
xc = ops16.to_channels_last16(torch.zeros(N, Cin, T, H, W).cuda())
plan = ops16.plan_for(xc, Conv(Cin, Cout, k, st, pd))
_, wt = ops16.conv_w_transform(plan, w.cuda(), need_wf=False)
for lo in range(0, Cout, 32):
    dy = torch.zeros(N, Cout, T, H, W)
    hi = min(lo + 32, Cout)
    dy[:, lo:hi] = bf(torch.randn(N, hi - lo, T, H, W, generator=g))
    x = torch.zeros(N, Cin, T, H, W, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv3d(x, w.double(), stride=st, padding=pd), x, dy.double())
    dx = ops16.conv_dgrad(plan, ops16.to_channels_last16(dy.cuda()), wt)
    got = ops16.from_channels_last16(dx, Cin).cpu().double()
    err = (got - want).abs()
    print(f"dY channels {lo}..{hi}: max err {float(err.max()):.4f} (ref max {float(want.abs().max()):.3f}); wrong positions {int((err > 1e-2).any(1).sum())} of {err[:,0].numel()}")
