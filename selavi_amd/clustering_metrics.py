"""Clustering evaluation of a dumped head-logit file (mirrors /root/reference/clustering_metrics.py:41-175).

Device side: the joint argmax softmax64(v) * softmax64(a) (slv_av_argmax, no N x K fp64 matrix is materialised) and
the K x K vote table of the Hungarian match (slv_contingency: one pass over N instead of the reference's K*K masked
sums, :47-52).  NMI / aNMI / ARI / entropy / the assignment itself stay on the host like in the reference (sklearn,
scipy), on label vectors only.
"""
import pickle

import numpy as np
import torch

from ._lib import C, ptr, stream


def joint_argmax(logits_v, logits_a):
    """argmax_k softmax64(v)_k * softmax64(a)_k per row (clustering_metrics.py:121-126,140-144) -> int64 N (device)."""
    lv = logits_v.detach().float().contiguous()
    la = logits_a.detach().float().contiguous()
    assert lv.shape == la.shape and lv.dim() == 2 and lv.is_cuda
    out = torch.empty(lv.shape[0], dtype=torch.int64, device=lv.device)
    C.slv_av_argmax(ptr(lv), ptr(la), lv.shape[0], lv.shape[1], ptr(out), stream())
    return out


def contingency(preds, targets, preds_k, targets_k):
    """num_correct[c1, c2] = #{preds == c1 and targets == c2} (clustering_metrics.py:47-52) -> int64 K1 x K2 (device)."""
    p = preds.to(torch.int64).contiguous()
    t = targets.to(torch.int64).contiguous()
    assert p.shape == t.shape and p.dim() == 1 and p.is_cuda and t.is_cuda
    counts = torch.empty((preds_k, targets_k), dtype=torch.int64, device=p.device)
    bad = torch.empty(1, dtype=torch.int32, device=p.device)
    C.slv_contingency(ptr(p), ptr(t), p.numel(), preds_k, targets_k, ptr(counts), ptr(bad), stream())
    if int(bad.item()):
        raise ValueError("a cluster id or a label is outside [0, k)")          # the reference asserts (:73)
    return counts


def _hungarian_match(flat_preds, flat_targets, preds_k, targets_k):
    """-> list of (out_c, gt_c) (clustering_metrics.py:41-66)."""
    from scipy.optimize import linear_sum_assignment
    assert isinstance(flat_preds, torch.Tensor) and isinstance(flat_targets, torch.Tensor)
    assert preds_k == targets_k                                               # one to one (:45)
    num_samples = flat_targets.shape[0]
    num_correct = contingency(flat_preds.cuda(), flat_targets.cuda(), preds_k, targets_k).cpu().numpy().astype(np.float64)
    match = linear_sum_assignment(num_samples - num_correct)
    return [(int(o), int(g)) for o, g in zip(match[0], match[1])]


def cluster_acc(match, preds, targets, num_k=309, verbose=1):
    """clustering_metrics.py:83-92 (+ _acc :69-80)."""
    preds = np.asarray(preds.cpu() if torch.is_tensor(preds) else preds)
    table = np.zeros(num_k, dtype=np.int64)                                   # clusters without a match map to 0 (:85)
    for pred_i, target_i in match:
        table[pred_i] = target_i
    reordered = torch.from_numpy(table[preds])
    targets = targets.cpu().to(torch.long)
    assert reordered.shape == targets.shape
    assert reordered.max() < num_k and targets.max() < num_k
    return int((reordered == targets).sum()) / float(reordered.shape[0])


def k_means(path="cluster_fit_PS_matrices_scratch_vgg_sound_train.pkl", ncentroids=512, use_all_heads=False, PS=None,
            verbose=True):
    """clustering_metrics.py:95-175.  `PS` may be passed directly instead of a pickle path; returns the metrics."""
    from scipy.stats import entropy
    from sklearn.metrics import adjusted_mutual_info_score, adjusted_rand_score, normalized_mutual_info_score
    if PS is None:
        PS = pickle.load(open(path, "rb"))
    true_labels = np.asarray(PS[1].cpu().numpy() if torch.is_tensor(PS[1]) else PS[1])
    say = print if verbose else (lambda *a, **k: None)
    res = {}
    if use_all_heads:
        best_nmi, best, res["nmi_per_head"] = 0, None, []
        for h in range(len(PS[0])):
            lab = joint_argmax(PS[0][h].cuda(), PS[2][h].cuda()).cpu().numpy()
            nmi = normalized_mutual_info_score(lab, true_labels, average_method="arithmetic")
            say(f"Head {h}: {nmi}")
            res["nmi_per_head"].append(nmi)
            if nmi > best_nmi:
                best_nmi, best = nmi, lab
        self_labels_np = best
    else:
        self_labels_np = joint_argmax(PS[0].cuda(), PS[2].cuda()).cpu().numpy()
    res["self_labels"] = self_labels_np
    res["nmi"] = normalized_mutual_info_score(self_labels_np, true_labels, average_method="arithmetic")
    res["anmi"] = adjusted_mutual_info_score(self_labels_np, true_labels, average_method="arithmetic")
    res["ari"] = adjusted_rand_score(self_labels_np, true_labels)
    say(f"NMI-tolabels: {res['nmi']}\naNMI-tolabels: {res['anmi']}\naRI-tolabels: {res['ari']}")
    purities, entropies = [], []
    for sk_label in np.unique(self_labels_np):
        counts = np.unique(true_labels[self_labels_np == sk_label], return_counts=True)[1]
        purities.append(max(counts) / sum(1.0 * counts))
        entropies.append(entropy(counts / sum(1.0 * counts)))
    res["entropy"], res["purity"] = float(np.mean(entropies)), float(np.mean(purities))
    say(f"Avg entropy: {res['entropy']}   avg purity: {res['purity']}")
    low = {n: a for a, n in enumerate(np.unique(true_labels))}
    tl = torch.tensor([low[n] for n in true_labels])
    sl = torch.tensor(self_labels_np)
    match = _hungarian_match(sl, tl, ncentroids, ncentroids)
    res["acc"] = cluster_acc(match, sl, tl, ncentroids)
    say(f"Number of unique classes: {len(low)}\nNumber of centroids: {ncentroids}\nClustering Acc: {res['acc'] * 100}%")
    return res
