"""Golden vectors for the input pipeline, produced by EXECUTING the reference's
datasets/video_transforms.clip_augmentation (it imports only math/numpy/torch) in the build container.

    python tests/golden/make_input_golden.py        # needs /root/reference; writes tests/golden/clip_aug.npz

np.random is seeded per case and the draws the reference makes (size; y, x offsets; flip) are recovered by
replaying the same generator calls in the same order (video_transforms.py:52,121-125,158).
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import input_ref                                                     # noqa: E402

spec = importlib.util.spec_from_file_location("ref_video_transforms", "/root/reference/datasets/video_transforms.py")
vt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(vt)

CASES = [  # seed, T, H, W, min_scale, max_scale, crop, spatial_idx
    (1, 4, 40, 52, 32, 40, 28, -1),
    (2, 3, 52, 40, 32, 40, 28, -1),          # portrait
    (3, 2, 45, 61, 30, 50, 24, -1),
    (4, 2, 36, 48, 36, 36, 30, -1),          # short side already == size: no resize
    (5, 2, 40, 52, 32, 32, 28, 0), (6, 2, 40, 52, 32, 32, 28, 1), (7, 2, 40, 52, 32, 32, 28, 2),
    (8, 2, 52, 40, 32, 32, 28, 3), (9, 2, 52, 40, 32, 32, 28, 4), (10, 2, 40, 52, 32, 32, 28, 5),
    (11, 8, 128, 171, 128, 160, 112, -1),    # the training shapes (opt.py:43: crop 112)
    (12, 4, 171, 128, 128, 160, 112, -1),    # portrait
    (13, 4, 128, 171, 128, 128, 112, 4),     # eval: centre crop + flip
]

out = {}
for seed, T, H, W, lo, hi, crop, sidx in CASES:
    g = np.random.RandomState(seed)
    frames = g.randint(0, 256, size=(T, H, W, 3)).astype(np.uint8)
    np.random.seed(seed)
    y = vt.clip_augmentation(torch.from_numpy(frames), spatial_idx=sidx, min_scale=lo, max_scale=hi, crop_size=crop)
    # replay the draws
    np.random.seed(seed)
    size = int(round(np.random.uniform(lo, hi)))
    nh, nw = input_ref.resized_shape(H, W, size)
    if sidx == -1:
        yo = int(np.random.randint(0, nh - crop)) if nh > crop else 0
        xo = int(np.random.randint(0, nw - crop)) if nw > crop else 0
        flip = bool(np.random.uniform() < 0.5)
    else:
        yo, xo = input_ref.uniform_crop_offsets(nh, nw, crop, {0: 0, 1: 1, 2: 2, 3: 0, 4: 1, 5: 2}[sidx])
        flip = sidx in (3, 4, 5)
    k = f"c{seed}"
    if T * crop * crop <= 20000:
        out[k + "_frames"] = frames           # large cases: the test regenerates them (RandomState is a frozen stream)
    out[k + "_params"] = np.array([lo, hi, crop, sidx, size, nh, nw, yo, xo, int(flip), seed, T, H, W], dtype=np.int64)
    if T * crop * crop > 20000:               # large case: keep a strided sample + checksum, not the whole clip
        yy = y.numpy()
        out[k + "_sample"] = yy[:, ::3, ::7, ::5].copy()
        out[k + "_sum"] = np.array([np.ascontiguousarray(yy).view(np.uint32).astype(np.uint64).sum()])   # exact, order-free
    else:
        out[k + "_out"] = y.numpy()
    ref = input_ref.clip_augmentation_ref(frames, (nh, nw), yo, xo, flip, crop)
    d = np.abs(ref - y.numpy()).max()
    print(k, tuple(y.shape), "size", size, (nh, nw), (yo, xo), flip, "oracle-vs-reference max abs", d,
          "bit-equal" if np.array_equal(ref, y.numpy()) else "")
np.savez_compressed(os.path.join(HERE, "clip_aug.npz"), **out)
print("wrote", os.path.join(HERE, "clip_aug.npz"), os.path.getsize(os.path.join(HERE, "clip_aug.npz")), "bytes")
