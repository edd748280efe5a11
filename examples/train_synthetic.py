"""End-to-end SeLaVi training on synthetic data with the MI355X-native hot path.

Follows the control flow of the reference's driver (/root/reference/main.py:102-302: model, SGD, SK schedule,
checkpoint restore, BN warm-up, per-iteration `cluster()` when the schedule says so, loss on the pseudo
labels, checkpoint per epoch), with its dataset / logging / SLURM plumbing replaced by the synthetic dataset of
selavi_amd.data.  Every device-side operation goes through libselavi_hip.so.

    python examples/train_synthetic.py --epochs 2 --dataset-size 256 --batch 8 --frames 8 --size 64
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py ...
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from selavi_amd import model as smodel, ops, optim, sk_utils, train, utils
from selavi_amd.data import SyntheticAVDataset


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--dataset-size", dest="n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch size")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--mel", type=int, nargs=2, default=(40, 100))
    ap.add_argument("--num-clusters", type=int, default=16)
    ap.add_argument("--headcount", type=int, default=2)
    ap.add_argument("--nopts", type=int, default=4)
    ap.add_argument("--schedulepower", type=float, default=1.5)
    ap.add_argument("--base-lr", type=float, default=1e-2)
    ap.add_argument("--wd", type=float, default=1e-5)
    ap.add_argument("--dump-path", default="")
    ap.add_argument("--bn-warmup", type=int, default=2)
    ap.add_argument("--seed", type=int, default=31, help="opt.py:152")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="bf16: the video trunk trains on the 16-bit MFMA path (main.py:151 --use_fp16)")
    ap.add_argument("--feature-pass", dest="feature_pass", choices=("fp32", "fp32_folded", "fp32x2", "bf16"), default="fp32",
                    help="arithmetic of the SK round's eval forward over the dataset (sk_utils.py:137-233)")
    a = ap.parse_args(argv)
    # what sk_utils.optimize_L_sk_gpu / cluster read from `args` (opt.py)
    a.distribution, a.dist, a.diff_dist_every, a.diff_dist_per_head = "default", None, False, True
    a.gauss_sd, a.lamb, a.ind_groups, a.match, a.shuffle_sk_pass = 0.1, 20, 1, False, False
    return a


def main(argv=None):
    args = parse(argv)
    args.rank = int(os.environ.get("RANK", "0"))
    args.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SELAVI_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    group = None
    if args.world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SELAVI_BENCH_DIST_BACKEND", "nccl"), rank=args.rank,
                                world_size=args.world_size)
        group = dist.group.WORLD
    torch.manual_seed(args.seed)                                            # opt.py:152
    np.random.seed(args.seed)
    ops.set_benchmark(True)                                                 # main.py:187

    dataset = SyntheticAVDataset(n=args.n, T=args.frames, S=args.size, F=args.mel[0], Tp=args.mel[1],
                                 n_classes=args.num_clusters)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=args.world_size, rank=args.rank)
    loader = torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch, drop_last=True)

    model = smodel.load_model(vid_base_arch="r2plus1d_18", aud_base_arch="resnet9", use_mlp=True,
                              num_classes=args.num_clusters, pretrained=False, norm_feat=False,
                              use_max_pool=False, headcount=args.headcount).cuda()          # main.py:105-114
    model.set_precision(args.precision)                                                     # :151-153
    optimizer = optim.SGD(model.parameters(), lr=args.base_lr, momentum=0.9, weight_decay=args.wd)   # :132-137
    net = model
    if args.world_size > 1:
        net = train.data_parallel(model, [local])                                           # :156-160

    n_dl, N = len(loader), len(dataset)
    selflabels = torch.zeros((N, args.headcount), dtype=torch.long, device="cuda")           # :166
    sk_schedule = train.sk_schedule(args.epochs, n_dl, args.nopts, args.schedulepower)       # :168-170
    start_epoch, sk_counter = 0, 0
    ckpt = os.path.join(args.dump_path, "checkpoint.pth.tar") if args.dump_path else ""
    if ckpt and os.path.exists(ckpt):                                                        # :174-197
        st = torch.load(ckpt, map_location="cuda", weights_only=False)
        net.load_state_dict(st["model"])
        optimizer.load_state_dict(st["optimizer"])
        start_epoch, selflabels, args.dist = st["epoch"], st["selflabels"].cuda(), st["dist"]
        include = [q / n_dl > start_epoch for q in sk_schedule]
        sk_counter = len(sk_schedule) - sum(include)
        sk_schedule = [q for q, inc in zip(sk_schedule, include) if inc]
    if start_epoch == 0 and args.bn_warmup:
        sampler.set_epoch(999)
        utils.warmup_batchnorm(args, net, loader, batches=args.bn_warmup, group=group)       # :199-201

    log = []
    for epoch in range(start_epoch, args.epochs):
        sampler.set_epoch(epoch)
        net.train()
        done = epoch * n_dl
        for it, (video, audio, _, selected, _) in enumerate(loader):                         # :263-302
            video, audio, selected = video.cuda(), audio.cuda(), selected.cuda()
            if done + it >= sk_schedule[-1]:
                with torch.no_grad():
                    sk_schedule.pop()
                    selflabels = sk_utils.cluster(args, selflabels, dataset, net, sk_counter, None, None, group,
                                                  (done + it) * args.batch * args.world_size)
                    sk_counter += 1
                net.train()
            loss = train.train_step(net, optimizer, video, audio, selflabels, selected, args.headcount)
            log.append(float(loss))
        if args.rank == 0:
            from sklearn.metrics.cluster import normalized_mutual_info_score
            nmi = normalized_mutual_info_score(selflabels[:, 0].cpu().numpy(), np.array(dataset._labels))
            print(f"epoch {epoch}: mean loss {np.mean(log[-n_dl:]):.4f}, SK rounds so far {sk_counter}, "
                  f"distinct labels head 0: {int(selflabels[:, 0].unique().numel())}, "
                  f"NMI(pseudo labels, synthetic classes) {nmi:.3f}", flush=True)
            if ckpt:
                torch.save({"epoch": epoch + 1, "dist": args.dist, "model": net.state_dict(),
                            "optimizer": optimizer.state_dict(), "selflabels": selflabels}, ckpt)   # :223-242
    if args.world_size > 1:
        dist.destroy_process_group()
    main.last_nmi = nmi if args.rank == 0 and args.epochs > start_epoch else None
    return log, selflabels, model


if __name__ == "__main__":
    main()
