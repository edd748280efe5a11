"""resnet18 / 34 / 50 audio trunks on the 16-bit path (trunk.precision = "bf16") against the fp32 CPU oracle: cosine of the
features and of every parameter gradient (train mode, blocks damped as in tests/test_audio_archs_gpu.py).  Informational:
`python tools/audio_archs_bf16.py` on a GPU box; what it printed is recorded in profiles/r05_notes.md."""
import sys
import traceback

import torch

sys.path.insert(0, ".")
from oracle import model_ref                                     # noqa: E402  (checker only)
from oracle.model_ref import portable_fill_, portable_init_      # noqa: E402
from selavi_amd import model as smodel                           # noqa: E402


def damp(trunk):
    with torch.no_grad():
        for li in range(1, 5):
            for blk in getattr(trunk, f"layer{li}"):
                (blk.bn3 if hasattr(blk, "bn3") else blk.bn2).weight.fill_(0.1)


def run(arch, B=8):
    spec = portable_fill_(torch.empty(B, 1, 128, 96), 6)
    out = {}
    for kind in ("bf16", "oracle32"):
        m = smodel.get_audio_feature_extractor(arch) if kind == "bf16" else model_ref.get_audio_feature_extractor(arch)
        portable_init_(m, seed=31)
        damp(m)
        w = portable_fill_(torch.empty(B, 2048 if arch == "resnet50" else 512), 9)
        if kind == "bf16":
            m = m.cuda().train()
            m.precision = "bf16"
            f = m(spec.cuda()).reshape(B, -1)
            (f.float() * w.cuda()).sum().backward()
            torch.cuda.synchronize()
        else:
            m = m.train()
            f = m(spec).reshape(B, -1)
            (f * w).sum().backward()
        out[kind] = (f.detach().double().cpu(), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()})
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    f, g = out["bf16"]
    f0, g0 = out["oracle32"]
    cs = {n: cos(g[n], g0[n]) for n in g0}
    worst = min(cs, key=cs.get)
    print(f"{arch} bf16: features cosine {cos(f, f0):.5f}; gradient cosines min {cs[worst]:.4f} ({worst}) "
          f"median {sorted(cs.values())[len(cs) // 2]:.5f}", flush=True)


for arch in ("resnet18", "resnet34", "resnet50"):
    try:
        run(arch)
    except Exception:
        print(f"{arch} bf16: FAILED", flush=True)
        traceback.print_exc()
