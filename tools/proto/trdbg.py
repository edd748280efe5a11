import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from selavi_amd import ops16
class Conv:
    def __init__(s, cin, cout, k, st, pd): s.in_channels, s.out_channels, s.kernel3, s.stride3, s.padding3 = cin, cout, k, st, pd
N, Cin, T, H, W, Cout = 2, 144, 5, 10, 10, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Cin, T, H, W, generator=g).bfloat16().float()
w = torch.randn(Cout, Cin, 3, 1, 1, generator=g) * 0.05
xc = ops16.to_channels_last16(x.cuda())
plan = ops16.plan_for(xc, Conv(Cin, Cout, (3,1,1), (1,1,1), (1,0,0)))
y, s1, s2 = ops16.conv_fwd(plan, xc, w.cuda(), want_stats=True)
got = ops16.from_channels_last16(y, Cout).cpu()
want = F.conv3d(x.double(), w.bfloat16().double(), padding=(1,0,0)).float()
bad = ~((got - want).abs() <= 0.02 * want.abs().max())
print("bad count", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()))
idx = bad.nonzero()
print("n", idx[:,0].unique().tolist(), "c", idx[:,1].unique().tolist()[:20], "t", idx[:,2].unique().tolist())
pix = (idx[:,3]*W + idx[:,4]).unique().tolist()
print("pixels", pix)
