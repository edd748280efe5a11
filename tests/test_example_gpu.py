"""The end-to-end driver (examples/train_synthetic.py: the reference's main.py control flow on synthetic data)
runs, clusters, checkpoints and resumes on one MI355X."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_synthetic_runs_clusters_and_resumes(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_synthetic as ts
    from selavi_amd import ops
    was = ops.benchmark
    try:
        _run(ts, tmp_path)
    finally:
        ops.benchmark = was          # the driver switches benchmark mode on (main.py:187); keep the suite hermetic


def _run(ts, tmp_path):
    common = ["--dataset-size", "64", "--batch", "8", "--frames", "4", "--size", "32", "--mel", "40", "36",
              "--num-clusters", "8", "--headcount", "2", "--nopts", "3", "--dump-path", str(tmp_path)]
    log, labels, model = ts.main(["--epochs", "2"] + common)
    assert len(log) == 2 * 8 and all(math.isfinite(v) for v in log)
    assert labels.shape == (64, 2) and int(labels.min()) >= 0 and int(labels.max()) < 8
    assert labels[:, 0].unique().numel() > 1            # an SK round ran (labels start as all-zero)
    ck = torch.load(os.path.join(tmp_path, "checkpoint.pth.tar"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "dist", "model", "optimizer", "selflabels"} and ck["epoch"] == 2   # main.py:223-230
    assert list(ck["model"].keys()) == list(model.state_dict().keys())
    # resume: one more epoch on top of the checkpoint
    log2, labels2, _ = ts.main(["--epochs", "3"] + common)
    assert len(log2) == 8 and all(math.isfinite(v) for v in log2)
    assert sum(log2) / len(log2) < sum(log[:8]) / 8     # still learning after the restore
