"""Eager vs HIP-graph replay of the training step at a small shape (BASELINE configs[0]: bs 4, 8 frames).

    python tools/graph_step.py [--batch 4 --frames 8] [--precision bf16 --batch 16 --frames 16 --mel 129 100 --K 309 --hc 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--mel", type=int, nargs=2, default=(40, 100))
    ap.add_argument("--K", type=int, default=28)
    ap.add_argument("--hc", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32")
    a = ap.parse_args()
    from selavi_amd import model as smodel, ops, optim, train
    ops.set_benchmark(True)
    dev = torch.device("cuda:0")
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(31)
        m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=a.K,
                              pretrained=False, norm_feat=False, use_max_pool=False, headcount=a.hc).to(dev).train()
        m.set_precision(a.precision)
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        g = torch.Generator(device=dev).manual_seed(1)
        video = torch.randn(a.batch, 3, a.frames, 112, 112, device=dev, generator=g)
        audio = torch.randn(a.batch, 1, a.mel[0], a.mel[1], device=dev, generator=g)
        labels = torch.randint(0, a.K, (3328, a.hc), device=dev, generator=g)
        sel = torch.randint(0, 3328, (a.batch,), device=dev, generator=g)
        if mode == "eager":
            for _ in range(4):
                train.train_step(m, opt, video, audio, labels, sel, a.hc)
            step = lambda: train.train_step(m, opt, video, audio, labels, sel, a.hc)
        else:
            gs = train.GraphedStep(m, opt, video, audio, labels, sel, a.hc, warmup=4)
            step = gs.replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        res[mode] = (ms, float(loss))
        print(f"{mode:6s} {ms:7.2f} ms/step  {a.batch / ms * 1e3:8.1f} clips/s  loss after {a.steps + 4 + (mode == 'graph')} steps {float(loss):.4f}", flush=True)


if __name__ == "__main__":
    main()
