"""Do two DDP ranks (gloo, sharing one GPU) stay bit-identical?  Bisects DDP options.

    python tests/diag/ddp_lockstep.py
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_ref                                         # noqa: E402
from oracle.model_ref import portable_fill_, portable_init_         # noqa: E402


def worker(rank, world, port, kw, ret, steps):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selavi_amd import model as smodel, optim, train
    torch.cuda.set_device(0)
    hc, K = 2, 7
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    m = m.cuda().train()
    net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], **kw)
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5)[rank * 2:(rank + 1) * 2].cuda()
    audio = portable_fill_(torch.empty(4, 1, 40, 36), 6)[rank * 2:(rank + 1) * 2].cuda()
    sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
    sel = torch.tensor([3, 17, 42, 63])[rank * 2:(rank + 1) * 2].cuda()
    out = []
    for s in range(steps):
        train.train_step(net, opt, video, audio, sl, sel, hc)
        torch.cuda.synchronize()
        if s == 0 and rank == 0:
            try:
                used = net.reducer._get_local_used_map().cpu().numpy()
                names = [k for k, _ in m.named_parameters()]
                print("local_used_map zeros:", int((used == 0).sum()), "of", used.size, [names[i] for i in np.nonzero(used == 0)[0][:4]], flush=True)
                print("ddp logging:", {k: v for k, v in net._get_ddp_logging_data().items() if "unused" in k or "static" in k}, flush=True)
            except Exception as e:
                print("no local used map:", e, flush=True)
        out.append({k: hashlib.sha256(v.detach().cpu().numpy().tobytes()).hexdigest()[:12] for k, v in m.state_dict().items()})
    ret[rank] = out
    dist.destroy_process_group()


def main():
    variants = {"default": {}, "find_unused": dict(find_unused_parameters=True), "nobcast": dict(broadcast_buffers=False),
                "view": dict(gradient_as_bucket_view=True), "static": dict(static_graph=True),
                "wrap_ddp": dict(broadcast_buffers=False, gradient_as_bucket_view=True),
                "static+find": dict(static_graph=True, find_unused_parameters=True),
                "all": dict(broadcast_buffers=False, gradient_as_bucket_view=True, static_graph=True)}
    only = sys.argv[1:] or list(variants)
    for i, name in enumerate(only):
        ret = mp.Manager().dict()
        mp.spawn(worker, args=(2, 29200 + i * 7 + os.getpid() % 100, variants[name], ret, 3), nprocs=2, join=True)
        for s in range(3):
            bad = [k for k in ret[0][s] if ret[0][s][k] != ret[1][s][k]]
            print(f"{name:12s} step {s}: {len(bad):3d} of {len(ret[0][s])} tensors differ  {bad[:3]}", flush=True)


if __name__ == "__main__":
    main()
