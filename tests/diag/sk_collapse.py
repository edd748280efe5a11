"""Why does a Sinkhorn-Knopp round return one distinct label?  Runs examples/train_synthetic.py's small configuration
with a printing logger and inspects the head logits that go into the third round.

    python tests/diag/sk_collapse.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from selavi_amd import sk_utils                                     # noqa: E402

orig = sk_utils.optimize_L_sk_gpu


def spy(args, PS, hc, logger=None, **kw):
    P0 = PS.clone()
    cost, L = orig(args, PS, hc, logger, **kw)
    col = P0.sum(0)
    print(f"  head {hc}: PS min {P0.min().item():.3e} max {P0.max().item():.3e} nan {int(torch.isnan(P0).sum())} "
          f"colsum min {col.min().item():.3e} max {col.max().item():.3e}  rows identical: "
          f"{bool((P0 - P0[0]).abs().max().item() == 0)}  cost {cost:.4f} distinct labels {int(L.unique().numel())}", flush=True)
    return cost, L


sk_utils.optimize_L_sk_gpu = spy
import train_synthetic                                              # noqa: E402


class Lg:
    def info(self, s, **k):
        if "error" in s or "Cost" in s:
            print("   log:", s, flush=True)


if __name__ == "__main__":
    orig_cluster = sk_utils.cluster

    def cluster(args, selflabels, dataset, model, sk_counter, logger, writer, group, iter_num):
        print(f"SK round at iter_num {iter_num}", flush=True)
        return orig_cluster(args, selflabels, dataset, model, sk_counter, Lg(), writer, group, iter_num)
    sk_utils.cluster = cluster
    train_synthetic.main(["--epochs", "3", "--dataset-size", "128", "--frames", "4", "--size", "32", "--mel", "40", "36",
                          "--num-clusters", "8", "--nopts", "4", "--batch", "16", "--headcount", "2"])
