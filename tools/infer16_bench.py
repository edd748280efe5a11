"""Eval-mode feature pass (the Sinkhorn-Knopp round's dominant cost): fp32 path vs the bf16 channels-last engine.

    python tools/infer16_bench.py [--batch 64]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--only16", action="store_true", help="skip the fp32 leg (for profiling)")
    a = ap.parse_args()
    from selavi_amd import infer16, model as smodel, ops
    ops.set_benchmark(True)
    dev = torch.device("cuda:0")
    torch.manual_seed(31)
    m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=309,
                          pretrained=False, norm_feat=False, use_max_pool=False, headcount=10).to(dev).eval()
    m.return_features = True
    g = torch.Generator(device=dev).manual_seed(1)
    video = torch.randn(a.batch, 3, a.frames, 112, 112, device=dev, generator=g)
    audio = torch.randn(a.batch, 1, 129, 100, device=dev, generator=g)
    eng = infer16.Engine(m)

    def timeit(fn, reps=5):
        with torch.no_grad():
            fn()
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ms32 = float('nan') if a.only16 else timeit(lambda: m(video, audio))
    ms16 = timeit(lambda: eng.features(video, audio))
    msv = timeit(lambda: eng.video_features(video))
    gf, mb = (81.04 * a.frames / 16 + 0.506) * a.batch, (518.1 * a.frames / 16 + 3.38) / 2 * a.batch   # bf16: half the bytes
    print(f"fp32 eval forward   B={a.batch}: {ms32:7.2f} ms  {a.batch / ms32 * 1e3:8.1f} clips/s")
    print(f"bf16 eval forward   B={a.batch}: {ms16:7.2f} ms  {a.batch / ms16 * 1e3:8.1f} clips/s  ({ms32 / ms16:.2f}x)  "
          f"{gf / ms16:.0f} TFLOP/s  {mb / ms16:.0f} GB/s algorithmic = {mb / ms16 / 8000:.3f} of the HBM roofline")
    print(f"  video trunk alone: {msv:7.2f} ms  {a.batch / msv * 1e3:8.1f} clips/s")


if __name__ == "__main__":
    main()
