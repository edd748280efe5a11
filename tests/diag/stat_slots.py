"""Do the statistics epilogues of the 16-bit convs write EVERY partial slot slv_cl16_conv_nblk announces?  (An unwritten
slot is summed by the finalize as whatever torch.empty left there.)  Fills the partial tensors with NaN first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from selavi_amd import ops16
from selavi_amd._lib import C, ptr, stream

class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd

def check(name, N, Cin, T, H, W, Cout, k, st, pd, stem=False):
    dev = torch.device("cuda")
    if stem:
        x = torch.randn(N, Cin, T, H, W, device=dev)
    else:
        x = ops16.to_channels_last16(torch.randn(N, Cin, T, H, W, device=dev))
    plan = ops16.plan_for(x, Conv(Cin, Cout, k, st, pd))
    w = torch.randn(Cout, Cin, *k, device=dev) * 0.05
    wf, _ = ops16.conv_w_transform(plan, w, need_wt=False)
    xin = ops16.stem_patch(plan, x) if stem else x
    y = torch.empty(plan.out_shape, dtype=torch.bfloat16, device=dev)
    ssum = torch.full((plan.Cout, plan.nblk), float("nan"), device=dev)
    ssq = torch.full_like(ssum, float("nan"))
    C.slv_cl16_conv(plan.g_fwd.ctypes.data, plan.mt_f, ptr(xin), ptr(wf), ptr(y), 0, 0, 0, 0, ptr(ssum), ptr(ssq), 0, 0, 0, 0, 0, 0, stream())
    torch.cuda.synchronize()
    bad = int(torch.isnan(ssum).sum()), int(torch.isnan(ssq).sum())
    print(f"{name:28s} nblk {plan.nblk:5d}  unwritten sum / sq slots: {bad}")

for N, F, Tp in ((4, 40, 36), (16, 129, 100)):
    H1, W1 = (F + 6 - 7) // 2 + 1, (Tp + 6 - 7) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    check(f"audio conv1 {N}x{F}x{Tp}", N, 1, 1, F, Tp, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), stem=True)
    check("audio layer1 3x3", N, 64, 1, H2, W2, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    check("audio layer2 3x3 s2", N, 64, 1, H2, W2, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    check("audio layer2 1x1 s2", N, 64, 1, H2, W2, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0))
    H3, W3 = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1
    check("audio layer2 3x3", N, 128, 1, H3, W3, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    check("audio layer3 3x3 s2", N, 128, 1, H3, W3, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    H4, W4 = (H3 - 1) // 2 + 1, (W3 - 1) // 2 + 1
    check("audio layer3 3x3", N, 256, 1, H4, W4, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    check("audio layer4 3x3 s2", N, 256, 1, H4, W4, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    H5, W5 = (H4 - 1) // 2 + 1, (W4 - 1) // 2 + 1
    check("audio layer4 3x3", N, 512, 1, H5, W5, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1))
check("video stem 4x3x4x32x32", 4, 3, 4, 32, 32, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), stem=True)
check("video l1 spatial small", 4, 64, 4, 16, 16, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))
check("video l1 temporal small", 4, 144, 4, 16, 16, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
