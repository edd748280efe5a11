"""Fixed cost of the data-parallel machinery on ONE GPU: the cfg2 step plain vs with the SyncBN all-reduces and the
DDP gradient buckets going through RCCL on a world of one rank (no wire time, every launch / event / stream hop).

    python tools/dist_overhead.py [--steps 10]
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--modes", default="", help="comma-separated subset of the modes")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29777")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from selavi_amd import model as smodel, ops, optim, train
    ops.set_benchmark(True)
    dev = torch.device("cuda:0")
    B, hc, K, N = a.batch, 10, 309, 170752
    g = torch.Generator(device=dev).manual_seed(1)
    video = torch.randn(B, 3, 16, 112, 112, device=dev, generator=g)
    audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)
    labels = torch.randint(0, K, (N, hc), device=dev, generator=g)
    sel = torch.randint(0, N, (B,), device=dev, generator=g)
    res = {}
    import warnings
    warnings.filterwarnings("ignore", message=".*AccumulateGrad node's stream.*")
    ddp_kw = {"ddp": {}, "ddp-nobcast": dict(broadcast_buffers=False),
              "ddp-nobcast-view": dict(broadcast_buffers=False, gradient_as_bucket_view=True),
              "ddp-nobcast-view-100MB": dict(broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=100),
              "ddp-nobcast-view-static": dict(broadcast_buffers=False, gradient_as_bucket_view=True, static_graph=True)}
    modes = ("plain", "syncbn", "syncbn+ddp", "syncbn+ddp-nobcast", "syncbn+ddp-nobcast-view", "native", "plain")
    if a.modes:
        modes = tuple(a.modes.split(","))
    for mode in modes:
        torch.manual_seed(31)
        m = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True, num_classes=K,
                              pretrained=False, norm_feat=False, use_max_pool=False, headcount=hc).to(dev).train()
        net = m
        m.set_sync_bn(mode != "plain")
        if mode == "native":
            net = train.data_parallel(m, [0], kind="native")
        elif "+" in mode:
            net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], **ddp_kw[mode.split("+")[1]])
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        for _ in range(4):
            train.train_step(net, opt, video, audio, labels, sel, hc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            train.train_step(net, opt, video, audio, labels, sel, hc)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        res.setdefault(mode, []).append(ms)
        print(f"{mode:32s} {ms:7.2f} ms/step  {B / ms * 1e3:7.1f} clips/s", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
