"""CPU checks of bench.py's roofline bookkeeping (no GPU): every kernel the cfg5 leg names is priced against the roof its
arithmetic intensity selects (ridge = 2 500 TFLOP/s dense bf16 MFMA / 8 TB/s HBM = 312.5 FLOP/B), and the
two-piece eval weights are exact sums of two bf16 pieces."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cfg5_roofline_objects_follow_the_arithmetic_intensity():
    import bench
    assert abs(bench.RIDGE_BF16 - 312.5) < 1e-9
    T = 32
    for name, want_bound, want_ai in (("l1_spatial", "mfma", 2 * 144 * 64 * 9 / ((64 + 144) * 2.0)),
                                      ("l1_temporal", "hbm", 2 * 144 * 64 * 3 / ((144 + 64) * 2.0))):
        iso = dict(ms=1.0, clips=64)
        live = dict(ms=2.0, clips=128, launches=8, min_ms=1.9, max_ms=2.1, how="test")
        frozen = dict(ms=2.1, launches=39, source="x", clips=128)
        r = bench._roof16(name, T, iso, live, frozen, pmc_bytes_16x16=1.0e6)
        assert r["bound"] == want_bound and abs(r["arithmetic_intensity_flop_per_byte"] - want_ai) < 1e-9
        assert (want_ai > bench.RIDGE_BF16) == (want_bound == "mfma")
        pos = 64 * T * 56 * 56
        L = bench.HOT16[name]
        gbs = pos * (L["cin"] + L["cout"]) * 2.0 / 1.0 / 1e6
        tf = pos * 2.0 * L["cin"] * L["cout"] * L["k"][0] * L["k"][1] * L["k"][2] / 1.0 / 1e9
        assert abs(r["hbm_gbs"] - gbs) < 1e-6 * gbs and abs(r["mfma_tflops"] - tf) < 1e-6 * tf
        # `frac` is the fraction of the BINDING roof, `peak` / `unit` name it; the other roof's fraction rides along
        if want_bound == "mfma":
            assert r["peak"] == 2500.0 and r["unit"] == "TFLOP/s" and abs(r["frac"] - tf / 2500.0) < 1e-12
        else:
            assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - gbs / 8000.0) < 1e-12
        assert r["in_step_live"]["clips_per_launch"] == 128 and r["in_step"]["clips_per_launch"] == 128
        assert abs(r["in_step_live"]["frac"] - r["frac"]) < 1e-9          # twice the clips in twice the time
        assert r["traffic"] == 1.0e6 * (64 * T) / 256.0 and r["algorithmic_bytes_per_launch"] == pos * (L["cin"] + L["cout"]) * 2.0


def test_two_piece_weights_are_exact_sums_of_two_bf16_pieces():
    """infer32._round16_: round to nearest even at 16 significand bits; afterwards the first two pieces of the three-piece split
    (csrc/igemm3.hpp: split3) represent the value exactly, the error is <= 2^-16 relative and unbiased."""
    from selavi_amd.infer32 import _round16_
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(200000, generator=g) * 0.05
    w0[:4] = torch.tensor([0.0, -0.0, 1.0, -1.5])
    w = _round16_(w0.clone())
    assert ((w.view(torch.int32) & 0xFF) == 0).all()
    h = (w.view(torch.int32) & -65536).view(torch.float32)
    r = w - h
    m = (r.view(torch.int32) & -65536).view(torch.float32)
    assert torch.equal(h + m, w) and (r - m).abs().max().item() == 0.0
    nz = w0 != 0
    rel = ((w - w0)[nz] / w0[nz].abs())
    assert rel.abs().max().item() <= 2.0 ** -16 and abs(rel.mean().item()) < 1e-7
    assert w[0].item() == 0.0 and w[2].item() == 1.0 and w[3].item() == -1.5
    # ties go to even: 1 + 2^-16 (exactly between 1 and 1 + 2^-15) -> 1; 1 + 3 * 2^-16 -> 1 + 2^-14
    t = _round16_(torch.tensor([1.0 + 2.0 ** -16, 1.0 + 3 * 2.0 ** -16]))
    assert t[0].item() == 1.0 and t[1].item() == 1.0 + 2.0 ** -14
