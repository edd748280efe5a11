"""get_spec on the GPU (mirrors /root/reference/datasets/audio_utils.py:14-74).

The reference slices one second of int16 PCM and calls python_speech_features.logfbank (0.6) per clip on the CPU.
Here the window start / volume draws are made on the host with the same np.random calls, and one kernel
(csrc/input.hip: slv_logfbank) computes the log mel filterbank energies of a whole batch in float64.
"""
import decimal
import functools

import numpy as np
import torch

from .._lib import C, ptr, stream

WINLEN, WINSTEP, NFFT, PREEMPH = 0.02, 0.01, 1024, 0.97          # audio_utils.py:47-64 + logfbank's default preemph


def _round_half_up(x):                                            # sigproc.round_half_up
    return int(decimal.Decimal(x).quantize(decimal.Decimal("1"), rounding=decimal.ROUND_HALF_UP))


@functools.lru_cache(maxsize=None)
def _tables(nfilt, nfft, samplerate, device):
    j = np.arange(nfft, dtype=np.float64)
    tw = np.concatenate([np.cos(2 * np.pi * j / nfft), np.sin(2 * np.pi * j / nfft)])
    hz2mel = lambda hz: 2595 * np.log10(1 + hz / 700.)            # base.hz2mel / mel2hz / get_filterbanks
    mel2hz = lambda mel: 700 * (10 ** (mel / 2595.0) - 1)
    melpoints = np.linspace(hz2mel(0), hz2mel(samplerate / 2), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate).astype(np.int32)
    assert bins.min() >= 0 and bins.max() <= nfft // 2 + 1 and (np.diff(bins) >= 0).all()
    return torch.from_numpy(tw).to(device), torch.from_numpy(bins).to(device)


def window(n_samples, fr_sec, num_sec=1, sample_rate=48000, use_temporal_jittering=False):
    """First sample of the clip's audio window (audio_utils.py:25-37), including the end-of-file clamp."""
    if use_temporal_jittering:
        fr_sec = fr_sec + np.random.uniform(-0.5, 0.5)
    fr_aud = int(np.round(fr_sec * sample_rate))
    to_aud = int(np.round(fr_sec * sample_rate) + sample_rate * num_sec)
    if fr_aud + (to_aud - fr_aud) > n_samples:
        fr_aud = n_samples - sample_rate * num_sec
    if fr_aud < 0:
        raise ValueError("the recording is shorter than the requested window")
    return fr_aud


def get_spec_batch(wav, starts, num_sec=1, sample_rate=48000, aud_spec_type=1, volumes=None, z_normalize=False):
    """wav: B x n int16 (device); starts: B first-sample indices; volumes: optional B factors.
    -> B x 1 x nfilt x frames float32 (frames = 99 for one second)."""
    assert wav.dtype == torch.int16 and wav.dim() == 2 and wav.is_cuda and wav.is_contiguous()
    B, n = wav.shape
    slen = sample_rate * num_sec
    starts = np.asarray(starts, dtype=np.int64)
    if (starts < 0).any() or (starts + slen > n).any():
        raise ValueError("audio window outside the recording")
    nfilt = 40 if aud_spec_type == 1 else 257
    frame_len, frame_step = _round_half_up(WINLEN * sample_rate), _round_half_up(WINSTEP * sample_rate)
    frames = C.slv_logfbank_frames(slen, frame_len, frame_step)
    tw, bins = _tables(nfilt, NFFT, sample_rate, str(wav.device))
    st = torch.from_numpy(starts).to(wav.device)
    vol = None if volumes is None else torch.as_tensor(np.asarray(volumes, dtype=np.float64)).to(wav.device)
    out = torch.empty((B, 1, nfilt, frames), dtype=torch.float32, device=wav.device)
    C.slv_logfbank(ptr(wav), ptr(st), ptr(vol), n, B, slen, frame_len, frame_step, NFFT, nfilt, ptr(tw), ptr(bins),
                   PREEMPH, int(bool(z_normalize)), ptr(out), stream())
    return out


def get_spec(wav, fr_sec, num_sec=1, sample_rate=48000, aug_audio=[], aud_spec_type=1, use_volume_jittering=False,
             use_temporal_jittering=False, z_normalize=False):
    """One clip, the reference's signature (:14-24): wav 1-D int16 (device) -> 1 x nfilt x frames float32."""
    start = window(wav.numel(), fr_sec, num_sec, sample_rate, use_temporal_jittering)
    vol = [np.random.uniform(0.9, 1.1)] if use_volume_jittering else None       # :42-43
    return get_spec_batch(wav.reshape(1, -1), [start], num_sec, sample_rate, aud_spec_type, vol, z_normalize)[0]
