"""World of ONE rank: gradients / weights after each step under DDP variants vs the plain step (must be bit-identical:
all-reduce over one rank is the identity and the kernels are deterministic).

    python tests/diag/ddp_static_grads.py [nccl|gloo]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_ref                                         # noqa: E402
from oracle.model_ref import portable_fill_, portable_init_         # noqa: E402


def run(kw, steps=3):
    from selavi_amd import model as smodel, optim, train
    hc, K = 2, 7
    m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc)
    portable_init_(m, seed=31)
    step_ref.set_dropout_p(m, 0.0)
    m = m.cuda().train()
    m.set_sync_bn(True)
    net = m if kw is None else torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], **kw)
    opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
    video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
    audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
    sl = torch.from_numpy((np.arange(64 * hc).reshape(64, hc) * 7919 % K).astype(np.int64)).cuda()
    sel = torch.tensor([3, 17, 42, 63]).cuda()
    out = []
    for s in range(steps):
        train.train_step(net, opt, video, audio, sl, sel, hc)
        torch.cuda.synchronize()
        out.append(({k: p.grad.detach().clone() for k, p in m.named_parameters()},
                    {k: p.detach().clone() for k, p in m.named_parameters()}))
    return out


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29788")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1)
    base = run(None)
    for name, kw in {"default": {}, "static": dict(static_graph=True),
                     "all": dict(broadcast_buffers=False, gradient_as_bucket_view=True, static_graph=True)}.items():
        got = run(kw)
        for s in range(3):
            badg = [(k, float((got[s][0][k] - base[s][0][k]).abs().max() / (base[s][0][k].abs().max() + 1e-30)))
                    for k in base[s][0] if not torch.equal(got[s][0][k], base[s][0][k])]
            badw = [k for k in base[s][1] if not torch.equal(got[s][1][k], base[s][1][k])]
            print(f"{backend} {name:8s} step {s}: grads differ {len(badg):3d}  weights differ {len(badw):3d}  {badg[:3]}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
