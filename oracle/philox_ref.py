"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
library) restated in numpy -- TEST INFRASTRUCTURE: the checker for slv_dropout_masks (csrc/heads.hip), which draws the
Dropout(0.3) masks of the MLP heads (/root/reference/model.py:79,85).  The reference leaves those draws to torch's CUDA
generator, whose stream is not reproducible outside torch; the product uses its own counter-based generator so that a
mask is a pure function of (seed, offset, element index).  Pinned by Random123's published known-answer vectors
(tests/test_oracle.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr: [n, 4] uint32, key: [2] uint32 -> [n, 4] uint32."""
    c = np.array(ctr, dtype=np.uint32).reshape(-1, 4).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(10):
        p0 = M0 * c[:, 0].astype(np.uint64)
        p1 = M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        with np.errstate(over="ignore"):
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c


def dropout_mask(seed, offset, p, n, first=0):
    """keep-mask (float32 0/1) of elements first .. first+n-1: element e uses word e % 4 of the block with counter
    (e // 4 lo, e // 4 hi, offset lo, offset hi) under key (seed lo, seed hi); keep iff u = word * 2^-32 >= p."""
    e = np.arange(first, first + n, dtype=np.uint64)
    blk = e >> np.uint64(2)
    ctr = np.stack([(blk & np.uint64(0xFFFFFFFF)).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32),
                    np.full(n, offset & 0xFFFFFFFF, dtype=np.uint32), np.full(n, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint32)], 1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    word = r[np.arange(n), (e & np.uint64(3)).astype(np.int64)]
    thresh = np.uint32(min(int(np.float32(p) * np.float32(4294967296.0)), 0xFFFFFFFF))
    return (word >= thresh).astype(np.float32)
