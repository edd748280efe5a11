"""Input pipeline: slv_clip_augment / slv_logfbank against the oracle and the golden vectors the
reference's own clip_augmentation produced (tests/golden/make_input_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import input_ref
from selavi_amd.datasets import audio_utils, video_transforms

GOLD = os.path.join(os.path.dirname(__file__), "golden", "clip_aug.npz")


def _bitsum(a):
    return int(np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).sum())


def _cases():
    d = np.load(GOLD)
    for k in sorted({n.split("_")[0] for n in d.files}, key=lambda s: int(s[1:])):
        lo, hi, crop, sidx, size, nh, nw, yo, xo, flip, seed, T, H, W = [int(v) for v in d[k + "_params"]]
        frames = d[k + "_frames"] if k + "_frames" in d.files else \
            np.random.RandomState(seed).randint(0, 256, size=(T, H, W, 3)).astype(np.uint8)
        yield k, d, frames, dict(lo=lo, hi=hi, crop=crop, sidx=sidx, nh=nh, nw=nw, yo=yo, xo=xo, flip=bool(flip), seed=seed)


# ---- CPU: the oracle and the host-side draws against the reference's outputs ----------------------------------------
def test_oracle_clip_augmentation_matches_reference_outputs():
    n_exact = 0
    for k, d, frames, p in _cases():
        ref = input_ref.clip_augmentation_ref(frames, (p["nh"], p["nw"]), p["yo"], p["xo"], p["flip"], p["crop"])
        if k + "_out" in d.files:
            # small images: torch takes a four-weight path there, <= 2 ulp from the production-size association
            assert np.abs(ref - d[k + "_out"]).max() <= 5e-7, k
            n_exact += int(np.array_equal(ref, d[k + "_out"]))
        else:                                                   # production sizes: bit-identical to the reference
            assert np.array_equal(ref[:, ::3, ::7, ::5], d[k + "_sample"]), k
            assert _bitsum(ref) == int(d[k + "_sum"][0]), k                # checksum of every output word
            n_exact += 1
    assert n_exact >= 4


def test_host_draws_follow_the_reference_generator_order():
    for k, d, frames, p in _cases():
        np.random.seed(p["seed"])
        got = video_transforms.sample_spatial_params(frames.shape[1], frames.shape[2], p["sidx"], p["lo"], p["hi"], p["crop"])
        assert got == (p["nh"], p["nw"], p["yo"], p["xo"], p["flip"]), (k, got, p)


def test_logfbank_oracle_shapes_and_filterbank():
    g = np.random.RandomState(0)
    wav = (g.randn(48000) * 3000).astype(np.int16)
    for t, nf in ((1, 40), (2, 257)):
        s = input_ref.get_spec_ref(wav, 0, aud_spec_type=t)
        assert s.shape == (1, nf, 99) and s.dtype == np.float32 and np.isfinite(s).all()
    fb = input_ref.get_filterbanks(40, 1024, 48000)
    assert fb.shape == (40, 513) and fb.min() >= 0 and fb.max() <= 1 and (fb.sum(1) > 0).all()
    # Parseval pins the power spectrum: sum_k |X_k|^2 over the full spectrum == nfft * sum x^2
    x = g.randn(960)
    X = np.fft.rfft(x, 1024)
    full = (np.abs(X) ** 2).sum() * 2 - np.abs(X[0]) ** 2 - np.abs(X[-1]) ** 2
    assert abs(full - 1024 * (x ** 2).sum()) < 1e-6 * full


# ---- GPU ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_clip_augment_bit_exact_vs_oracle_and_reference():
    for k, d, frames, p in _cases():
        prm = (p["nh"], p["nw"], p["yo"], p["xo"], p["flip"])
        y = video_transforms.clip_augmentation_batch([torch.from_numpy(frames).cuda()], [prm], p["crop"])[0].cpu().numpy()
        ref = input_ref.clip_augmentation_ref(frames, (p["nh"], p["nw"]), p["yo"], p["xo"], p["flip"], p["crop"])
        assert np.array_equal(y, ref), (k, np.abs(y - ref).max())
        if k + "_out" in d.files:
            assert np.abs(y - d[k + "_out"]).max() <= 5e-7, k
        else:
            assert np.array_equal(y[:, ::3, ::7, ::5], d[k + "_sample"]) and _bitsum(y) == int(d[k + "_sum"][0]), k


@pytest.mark.gpu
def test_clip_augment_reference_signature_and_ragged_batch():
    g = np.random.RandomState(5)
    clips = [g.randint(0, 256, size=(4, h, w, 3)).astype(np.uint8) for h, w in ((128, 171), (171, 128), (140, 140), (128, 228))]
    np.random.seed(77)
    prms = [video_transforms.sample_spatial_params(c.shape[1], c.shape[2], -1, 128, 160, 112) for c in clips]
    y = video_transforms.clip_augmentation_batch([torch.from_numpy(c).cuda() for c in clips], prms, 112).cpu().numpy()
    assert y.shape == (4, 3, 4, 112, 112)
    for b, (c, (nh, nw, yo, xo, fl)) in enumerate(zip(clips, prms)):
        assert np.array_equal(y[b], input_ref.clip_augmentation_ref(c, (nh, nw), yo, xo, fl, 112)), b
    # the single-clip entry point draws from np.random exactly like the reference
    np.random.seed(77)
    one = video_transforms.clip_augmentation(torch.from_numpy(clips[0]).cuda(), -1, 128, 160, 112).cpu().numpy()
    assert np.array_equal(one, y[0])
    with pytest.raises(NotImplementedError):
        video_transforms.clip_augmentation(torch.from_numpy(clips[0]).cuda(), colorjitter=True)
    with pytest.raises(ValueError):
        video_transforms.clip_augmentation_batch([torch.from_numpy(clips[0]).cuda()], [(128, 171, 100, 0, False)], 112)


@pytest.mark.gpu
@pytest.mark.parametrize("spec_type", [1, 2])
def test_logfbank_matches_numpy_restatement(spec_type):
    g = np.random.RandomState(3)
    n = 48000 * 3
    t = np.arange(n) / 48000.0
    wavs = np.stack([
        (g.randn(n) * 2500).clip(-32768, 32767),                                          # noise
        8000 * np.sin(2 * np.pi * 440 * t) + 500 * np.sin(2 * np.pi * 9000 * t) + g.randn(n) * 20,   # tones + floor
        np.where((t > 1.2) & (t < 1.5), g.randn(n) * 9000, 0.0),                           # digital silence + burst
    ]).astype(np.int16)
    starts = [0, 48000 + 123, 48000]
    vols = [1.0, 0.93, 1.07]
    for z in (False, True):
        out = audio_utils.get_spec_batch(torch.from_numpy(wavs).cuda(), starts, aud_spec_type=spec_type,
                                         volumes=vols, z_normalize=z).cpu().numpy()
        for b in range(3):
            ref = input_ref.get_spec_ref(wavs[b], starts[b], aud_spec_type=spec_type, volume=vols[b], z_normalize=z)
            assert out[b].shape == ref.shape
            # float64 DFT on both sides: agreement to a few float32 ulp of log-energies up to ~30
            assert np.abs(out[b] - ref).max() <= 2e-5, (b, z, np.abs(out[b] - ref).max())
            empty = ref == ref.min()
            assert np.array_equal(out[b][empty], ref[empty])                               # log(eps) cells are exact
    # no volume factor: the int16 samples are used as they are
    o = audio_utils.get_spec_batch(torch.from_numpy(wavs).cuda(), starts, aud_spec_type=spec_type).cpu().numpy()
    r = input_ref.get_spec_ref(wavs[0], 0, aud_spec_type=spec_type)
    assert np.abs(o[0] - r).max() <= 2e-5


@pytest.mark.gpu
def test_get_spec_reference_signature_and_window_clamp():
    g = np.random.RandomState(4)
    wav = (g.randn(48000 * 2) * 1000).astype(np.int16)
    np.random.seed(9)
    s = audio_utils.get_spec(torch.from_numpy(wav).cuda(), 1.4, use_volume_jittering=True).cpu().numpy()   # clamps to the last second
    np.random.seed(9)
    v = np.random.uniform(0.9, 1.1)
    ref = input_ref.get_spec_ref(wav, int(np.round(1.4 * 48000)), volume=v)
    assert s.shape == (1, 40, 99) and np.abs(s - ref).max() <= 2e-5
    with pytest.raises(ValueError):
        audio_utils.get_spec_batch(torch.from_numpy(wav[None]).cuda(), [48000 * 2 - 10])
