"""Forward of the column kernel at the step test's shapes against torch (diagnostic)."""
import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from selavi_amd import ops16
from tests.test_train16_gpu import _bf, _cl, _ncthw, _Conv
for case in [(8, 45, 8, 32, 32, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)), (8, 144, 8, 32, 32, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
             (8, 64, 8, 32, 32, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))]:
    N, Cin, T, H, W, Cout, k, st, pd = case
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(N, Cin, T, H, W, generator=g))
    w = torch.randn(Cout, Cin, *k, generator=g) * (Cin * 3) ** -0.5
    ss = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3]).contiguous()
    xc = _cl(x)
    plan = ops16.plan_for(xc, _Conv(Cin, Cout, k, st, pd))
    xa = _bf(torch.addcmul(ss[1].view(1, -1, 1, 1, 1), x, ss[0].view(1, -1, 1, 1, 1)).clamp_min(0))
    want = F.conv3d(xa, _bf(w), stride=st, padding=pd)
    y, s1, s2 = ops16.conv_fwd(plan, xc, w.cuda(), in_ss=ss.cuda(), in_relu=True, want_stats=True)
    got = _ncthw(y, Cout)
    err = (got - want).abs()
    print(case[:6], "nblk", plan.nblk, "max err %.4f of %.3f" % (float(err.max()), float(want.abs().max())),
          "bad frac %.5f" % float((err > 0.02 * want.abs().max()).float().mean()),
          "stat rel err %.2e" % float((s1.double().sum(1).cpu() - got.double().sum((0, 2, 3, 4))).abs().max() / got.double().sum((0, 2, 3, 4)).abs().max()))
    g2 = got.double()
    print("   sumsq rel err %.2e" % float((s2.double().sum(1).cpu() - (g2 * g2).sum((0, 2, 3, 4))).abs().max() / (g2 * g2).sum((0, 2, 3, 4)).abs().max()),
          " output hash", float(y.float().sum()), float(y.float().abs().sum()), " one-ulp flips vs torch:", int((err > 0).sum()), "of", err.numel())
    bad = (err > 0.02 * want.abs().max())
    if bad.any():
        idx = bad.nonzero()
        print("  first bad (n,c,t,h,w):", idx[:5].tolist(), " per-n:", bad.sum((1, 2, 3, 4)).tolist(), " per-t:", bad.sum((0, 1, 3, 4)).tolist())
