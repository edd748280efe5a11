"""Golden vectors for the evaluation metrics, produced by EXECUTING the reference's clustering_metrics.k_means
(and its _hungarian_match / cluster_acc) in the build container on seeded synthetic head logits.

    python tests/golden/make_eval_golden.py         # needs /root/reference; writes tests/golden/eval_metrics.npz
"""
import contextlib
import importlib.util
import io
import os
import pickle
import re
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_clustering_metrics", "/root/reference/clustering_metrics.py")
cm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cm)


def synth(seed, N, K, heads, n_classes):
    g = np.random.RandomState(seed)
    labels = g.randint(0, n_classes, size=N)
    proto_v, proto_a = g.randn(n_classes, K), g.randn(n_classes, K)
    v = [(1.5 * proto_v[labels] + g.randn(N, K)).astype(np.float32) for _ in range(heads)]
    a = [(1.5 * proto_a[labels] + g.randn(N, K)).astype(np.float32) for _ in range(heads)]
    return v, labels * 3 + 1, a                       # sparse label ids: exercises translate_to_low_classes (:160-161)


out = {}
for name, (seed, N, K, heads, ncls) in {"a": (1, 3000, 16, 3, 16), "b": (2, 5000, 28, 1, 20)}.items():
    v, labels, a = synth(seed, N, K, heads, ncls)
    all_heads = heads > 1
    PS = [[torch.from_numpy(x) for x in v], torch.from_numpy(labels), [torch.from_numpy(x) for x in a]] if all_heads \
        else [torch.from_numpy(v[0]), torch.from_numpy(labels), torch.from_numpy(a[0])]
    with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as f:
        pickle.dump(PS, f)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cm.k_means(path=f.name, ncentroids=K, use_all_heads=all_heads)
    os.unlink(f.name)
    txt = buf.getvalue()
    print(txt)
    num = lambda pat: float(re.search(pat + r"\s*([-+0-9.eE]+)", txt).group(1))
    out[name + "_cfg"] = np.array([seed, N, K, heads, ncls])
    out[name + "_metrics"] = np.array([num("NMI-tolabels:"), num("aNMI-tolabels:"), num("aRI-tolabels:"),
                                       num("Avg entropy:"), num("avg purity:"), num(r"Clustering Acc:") / 100.0])
    if all_heads:
        out[name + "_head_nmi"] = np.array([float(x) for x in re.findall(r"Head \d+: ([-+0-9.eE]+)", txt)])
np.savez(os.path.join(HERE, "eval_metrics.npz"), **out)
print({k: v for k, v in out.items()})
