"""Ablations of the 8-wave conv kernel (csrc/conv_cl16_g8.hip, -DSLV_G8_ABL=n / -DSLV_G8_REQ_IN_M=k variants built by
tools/g8_ablate.sh into tools/proto/libselavi_g8_<tag>.so): the plain forward (no prologue, no statistics) of two layer-2
shapes per variant, each variant in its own process.  Results of the ablated variants are wrong by construction.
Usage: python tools/g8_ablate.py            (parent: runs every variant)      python tools/g8_ablate.py --one   (child)"""
import glob, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--one" not in sys.argv:
    libs = [("shipped", None)] + sorted((os.path.basename(p)[len("libselavi_g8_"):-3], p) for p in glob.glob(os.path.join(ROOT, "tools/proto/libselavi_g8_*.so")))
    for tag, path in libs:
        env = dict(os.environ)
        if path:
            env["SELAVI_HIP_LIB"] = path
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True, timeout=300)
        print(f"{tag:12s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
    sys.exit(0)
import torch
from selavi_amd import ops16
from selavi_amd._lib import C


class Conv:
    def __init__(self, cin, cout, k, st, pd):
        self.in_channels, self.out_channels, self.kernel3, self.stride3, self.padding3 = cin, cout, k, st, pd


def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
out = []
C.slv_cl16_g8_mode(2)
for name, Cin, T, H, W, Cout, k, st, pd in (("l2.1.spatial", 128, 8, 28, 28, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                            ("l2.1.temporal", 288, 8, 28, 28, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
                                            ("l3.1.spatial", 256, 4, 14, 14, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                            ("l3.1.temporal", 576, 4, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0))):
    B = 64
    x = ops16.to_channels_last16(torch.randn(B, Cin, T, H, W, device=dev, generator=g))
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) * 0.05
    ss = torch.stack([torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.1]).contiguous()
    plan = ops16.plan_for(x, Conv(Cin, Cout, k, st, pd))
    wf, wt = ops16.conv_w_transform(plan, w)
    t_plain = timeit(lambda: ops16.conv_fwd(plan, x, w, want_stats=False, wf=wf))
    t_full = timeit(lambda: ops16.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf))
    dy = torch.randn(plan.out_shape, device=dev, generator=g).to(torch.bfloat16)
    dy[..., Cout:] = 0
    t_dg = timeit(lambda: ops16.conv_dgrad(plan, dy, wt))
    flop = 2.0 * B * plan.out_dims[0] * plan.out_dims[1] * plan.out_dims[2] * Cout * Cin * k[0] * k[1] * k[2]
    out.append(f"{name}: plain {t_plain:.3f} ms ({flop / t_plain / 1e9:.0f} TF) train {t_full:.3f} dgrad {t_dg:.3f} ms")
print(" | ".join(out))
