"""Seeded synthetic inputs shared by tests, smoke() and bench.py (no reference needed)."""
import numpy as np
import torch

from oracle.model_ref import portable_fill_
from oracle.sk_ref import softmax64


def synth_PS(N, K, scale, seed):
    """PS = softmax64(s*G1) * softmax64(s*G2) with portable N(0,1) G."""
    g1 = portable_fill_(torch.empty(N, K, dtype=torch.float64), seed, kind="normal").numpy()
    g2 = portable_fill_(torch.empty(N, K, dtype=torch.float64), seed + 1, kind="normal").numpy()
    return softmax64(scale * g1) * softmax64(scale * g2)


def synth_logits(N, K, scale, seed):
    g1 = portable_fill_(torch.empty(N, K, dtype=torch.float32), seed, kind="normal").numpy()
    g2 = portable_fill_(torch.empty(N, K, dtype=torch.float32), seed + 1, kind="normal").numpy()
    return (scale * g1).astype(np.float32), (scale * g2).astype(np.float32)
