"""Full training step at large per-GPU batches (BASELINE configs[4]: 128 clips x 32 frames per GPU), fp32.

    python tools/large_batch.py --batch 128 --frames 32 [--benchmark]

Tensors beyond the 32-bit buffer range are convolved in batch slices (ops.ConvPlan.chunks); everything stays
resident in HBM (288 GB).  Prints clips/s, ms/step, algorithmic TFLOP/s and the peak allocation."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--benchmark", action="store_true")
    a = ap.parse_args()
    from selavi_amd import model as smodel, ops, optim, train
    ops.set_benchmark(a.benchmark)
    dev = torch.device("cuda:0")
    B, T, hc, K, N = a.batch, a.frames, 10, 309, 170752
    torch.manual_seed(31)
    m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=K,
                          pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).to(dev)
    m.train()
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(1)
    video = torch.randn(B, 3, T, 112, 112, device=dev, generator=g)
    audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)
    labels = torch.randint(0, K, (N, hc), device=dev, generator=g)
    sel = torch.randint(0, N, (B,), device=dev, generator=g)
    losses = [float(train.train_step(m, opt, video, audio, labels, sel, hc))]      # plan-building pass
    losses.append(float(train.train_step(m, opt, video, audio, labels, sel, hc)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train.train_step(m, opt, video, audio, labels, sel, hc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    losses.append(float(loss))
    gflop = 3 * (81.04 * T / 16 + 0.506 + 0.0168) * B
    sliced = sum(p.chunks is not None for p in ops.ConvPlan._cache.values())
    print(f"B={B} T={T} fp32: {dt * 1e3:.1f} ms/step  {B / dt:.1f} clips/s  {gflop / dt / 1e3:.1f} TFLOP/s algorithmic "
          f"({gflop / dt / 1e3 / 157.3:.3f} of fp32 MFMA peak)  peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  "
          f"sliced conv plans {sliced}/{len(ops.ConvPlan._cache)}  losses {['%.4f' % l for l in losses]}", flush=True)


if __name__ == "__main__":
    main()
