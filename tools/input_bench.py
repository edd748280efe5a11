"""Throughput of the device input pipeline (slv_clip_augment, slv_logfbank) at the cfg2 batch shape.

    python tools/input_bench.py            # on the GPU box
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selavi_amd.datasets import audio_utils, video_transforms  # noqa: E402


def timeit(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B, T, H, W, S = 16, 16, 128, 171, 112
    g = torch.Generator(device="cuda").manual_seed(0)
    clips = torch.randint(0, 256, (B, T, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    np.random.seed(0)
    prms = [video_transforms.sample_spatial_params(H, W, -1, 128, 160, S) for _ in range(B)]
    out = torch.empty((B, 3, T, S, S), device="cuda")
    ms = timeit(lambda: video_transforms.clip_augmentation_batch(clips, prms, S, out=out))
    src_bytes = sum(T * 3 * (S * H / nh) * (S * W / nw) for nh, nw, *_ in prms)      # source pixels under the crop
    byt = src_bytes + out.numel() * 4
    print(f"clip_augment  B={B} T={T} {H}x{W}->{S}: {ms * 1e3:8.1f} us  {B / ms * 1e3:10.0f} clips/s  "
          f"{byt / ms / 1e6:7.1f} GB/s (algorithmic: crop footprint read + clip written)")
    wav = (torch.randn(B, 48000 * 2, device="cuda", generator=g) * 3000).to(torch.int16)
    for t in (1, 2):
        ms = timeit(lambda: audio_utils.get_spec_batch(wav, [100] * B, aud_spec_type=t))
        flop = B * 99 * 513 * 960 * 4
        print(f"logfbank type {t} B={B}: {ms * 1e3:8.1f} us  {B / ms * 1e3:10.0f} clips/s  {flop / ms / 1e9:6.2f} TFLOP/s fp64 (direct DFT)")


if __name__ == "__main__":
    main()
