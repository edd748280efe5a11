import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from selavi_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
for geo in [(2, 256, 2, 4, 4, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (2, 576, 2, 4, 4, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
            (2, 128, 4, 8, 8, 460, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (2, 460, 4, 4, 4, 256, (3, 1, 1), (2, 1, 1), (1, 0, 0))]:
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    x = torch.randn(Bn, Cin, T, H, W); w = torch.randn(Cout, Cin, *k) * 0.05
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, stride=st, padding=pd)
    dy = torch.randn(y_ref.shape)
    y_ref.backward(dy.double())
    plan = ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)
    wt = ops.conv_wt_transform(plan, w.to(dev))
    def rel(a, b): return ((a.cpu().double() - b).norm() / b.norm()).item()
    for op in range(3):
        for cfg in [0] + plan.candidates(op):
            c = [0, 0, 0]; c[op] = cfg; plan.set_configs(*c)
            if op == 0: e = rel(ops.conv_fwd(plan, x.to(dev), w.to(dev))[0], y_ref.detach())
            elif op == 1: e = rel(ops.conv_dgrad(plan, dy.to(dev), wt), xr.grad)
            else: e = rel(ops.conv_wgrad(plan, dy.to(dev), x.to(dev)).view_as(w), wr.grad)
            print(geo[1], geo[5], "op", op, "cfg mt=%d nt=%d sp=%d" % (cfg & 255, (cfg >> 8) & 255, cfg >> 16), "rel err %.2e" % e)
