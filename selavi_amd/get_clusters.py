"""Dump head logits of a dataset for offline clustering metrics (mirrors /root/reference/get_clusters.py:57-168).

Each rank forwards its contiguous slice of the dataset in eval mode (GAP features when headcount > 1, like the
reference :92-94), the slices are all-gathered, and rank 0 applies every head to the feature bank and pickles
[PS_v_heads, labels, PS_a_heads] -- the file format clustering_metrics.k_means reads.
"""
import os
import pickle

import torch
import torch.distributed as dist


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


@torch.no_grad()
def get_cluster_assignments_gpu(args, dataset, model, logger=None, device="cuda", group=None):
    model.eval()
    m = _unwrap(model)
    N = len(dataset)
    W, rank = getattr(args, "world_size", 1), getattr(args, "rank", 0)
    local = N // W
    indices = list(range(rank * local, (rank + 1) * local))                   # :70-76
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, sampler=indices,
                                         num_workers=getattr(args, "workers", 0), shuffle=False)
    hc = args.headcount
    if hc > 1:
        m.return_features = True                                              # :92-94
    fv, fa, lab = [], [], []
    engine16 = None                                                           # opt-in bf16 forward (infer16.py), as in cluster()
    if (getattr(args, "feature_pass", None) or os.environ.get("SELAVI_FEATURE_PASS", "fp32")) == "bf16":
        from . import infer16
        engine16 = infer16.Engine(m)
    try:
        for batch in loader:
            video, audio, label = batch[0].cuda(non_blocking=True), batch[1].cuda(non_blocking=True), batch[2].cuda()
            if engine16 is None:
                v, a = model(video, audio)
            else:
                v, a = engine16.features(video, audio)
                if hc == 1:
                    v, a = m.mlp_v.forward(v), m.mlp_a.forward(a)
                    if m.norm_feat:
                        v, a = torch.nn.functional.normalize(v, p=2, dim=1), torch.nn.functional.normalize(a, p=2, dim=1)
            if hc == 1:
                v, a = v.double(), a.double()                                 # the reference's DoubleTensor bank (:95-96)
            fv.append(v), fa.append(a), lab.append(label.long())
    finally:
        if hc > 1:
            m.return_features = False
    fv, fa, lab = torch.cat(fv), torch.cat(fa), torch.cat(lab)
    if W > 1:                                                                  # :110-119, once per slice not per batch
        def gather(t):
            parts = [torch.empty_like(t) for _ in range(W)]
            dist.all_gather(parts, t.contiguous(), group=group)
            return torch.cat(parts)
        fv, fa, lab = gather(fv), gather(fa), gather(lab)
    out = None
    if rank == 0:                                                              # :146-163
        if hc > 1:
            PS_v = [getattr(m, f"mlp_v{h}").forward(fv) for h in range(hc)]
            PS_a = [getattr(m, f"mlp_a{h}").forward(fa) for h in range(hc)]
        else:
            PS_v, PS_a = fv, fa
        out = [PS_v, lab.cpu(), PS_a]
        if getattr(args, "output_dir", None):
            os.makedirs(args.output_dir, exist_ok=True)
            with open(os.path.join(args.output_dir, f"{args.exp_desc}.pkl"), "wb") as handle:
                pickle.dump(out, handle, protocol=pickle.HIGHEST_PROTOCOL)
    if W > 1:
        dist.barrier(group=group)
    return out
