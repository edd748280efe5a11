"""Layer-1 temporal weight gradient (144 -> 64, (3,1,1)) at B clips x 16 frames x 56 x 56: the plain column-order kernel + the
BatchNorm-backward reduce pass it is followed by, against conv_wgrad(bnr=...) (csrc/wgrad_cl16_t2.hip / wgrad_cl16_tacc.hip:
both in one).  Usage: python tools/wgrad_bnr_bench.py [batch] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import ops16

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
class Conv:
    in_channels, out_channels, kernel3, stride3, padding3 = 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)
y = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
plan = ops16.plan_for(y, Conv)
dy = ops16.to_channels_last16(torch.randn(B, 64, 16, 56, 56, device=dev, generator=g))
gm = ops16.to_channels_last16(torch.randn(B, 144, 16, 56, 56, device=dev, generator=g))
ss = torch.stack([torch.rand(144, device=dev, generator=g) + 0.5, torch.randn(144, device=dev, generator=g) * 0.1]).contiguous()
mi = torch.stack([torch.randn(144, device=dev, generator=g) * 0.1, torch.rand(144, device=dev, generator=g) + 0.5]).contiguous()
w = torch.randn(64, 144, 3, 1, 1, device=dev, generator=g) * 0.05
gamma = torch.ones(144, device=dev)

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

t_plain = timeit(lambda: ops16.conv_wgrad(plan, dy, y, in_ss=ss, in_relu=True))
t_red = timeit(lambda: ops16.bn_bwd(gm, y, mi, gamma, ss_mask=ss))
t_bnr = timeit(lambda: ops16.conv_wgrad(plan, dy, y, in_ss=ss, in_relu=True, bnr=(mi, w))) if ops16.wgrad_bnr_available(plan) else float("nan")
nb = (y.numel() + dy.numel()) * 2
print(f"B={B}: plain weight gradient {t_plain:.3f} ms + reduce pass (incl. its finish) {t_red:.3f} ms = {t_plain + t_red:.3f} ms;  "
      f"weight gradient with the sums {t_bnr:.3f} ms ({nb / t_bnr / 1e6:.0f} GB/s algorithmic)")
