"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's input pipeline for one clip:

  clip_augmentation_ref   /root/reference/datasets/video_transforms.py:462-510 (normalise, THWC->TCHW,
                          spatial_sampling :420-459 = short-side bilinear resize :35-80, crop :101-134 / :167-210,
                          horizontal flip :137-164, TCHW->CTHW) with the random draws passed in explicitly.
                          Pinned against the reference itself (tests/golden/make_input_golden.py executes the
                          reference's clip_augmentation in the build container).
  logfbank_ref            /root/reference/datasets/audio_utils.py:46-72, which calls python_speech_features.logfbank
                          (third-party, pinned ==0.6 in environment.yml:145, ABSENT from /root/reference and from this
                          image).  The function below restates that package's published algorithm (base.py: fbank,
                          get_filterbanks, hz2mel, mel2hz; sigproc.py: preemphasis, framesig, powspec) in numpy float64.
                          PARITY UNPINNED by reference outputs: nothing in the reference holds a vector for it; the
                          arithmetic oracle underneath is numpy.fft.rfft.
"""
import decimal
import math

import numpy as np

MEAN = (0.45, 0.45, 0.45)                    # video_transforms.py:13-14
STD = (0.225, 0.225, 0.225)


# ---- video ---------------------------------------------------------------------------------------------------------
def resized_shape(H, W, size):
    """random_short_side_scale_jitter's output shape for a drawn `size` (video_transforms.py:52-67)."""
    if (W <= H and W == size) or (H <= W and H == size):
        return H, W
    if W < H:
        return int(math.floor((float(H) / W) * size)), size
    return size, int(math.floor((float(W) / H) * size))


def _fma(a, b, c):
    """float32 fused multiply-add: the float64 product of two float32 values is exact, so one final rounding."""
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)


def _axis(n_in, n_out):
    """torch's bilinear source indices/weights, align_corners=False, float32 arithmetic (aten UpSample.h:
    area_pixel_compute_source_index + guard_index_and_lambda).  The torch CPU build evaluates
    scale*(dst+0.5)-0.5 as ONE fused multiply-add (found by bit-comparison with the reference's outputs;
    with separate roundings the weights are off by up to 1e-5 at large indices)."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = _fma(scale, dst + np.float32(0.5), np.float32(-0.5))
    src = np.maximum(src, np.float32(0))
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    l1 = np.clip(src - i0.astype(np.float32), np.float32(0), np.float32(1)).astype(np.float32)
    i1 = i0 + (i0 < n_in - 1)
    return i0, i1, (np.float32(1) - l1).astype(np.float32), l1


def clip_augmentation_ref(frames_u8, new_hw, y_off, x_off, flip, crop):
    """frames_u8 T x H x W x 3 uint8 -> 3 x T x crop x crop float32."""
    f = frames_u8.astype(np.float32) / np.float32(255.0)
    f = (f - np.array(MEAN, dtype=np.float32)) / np.array(STD, dtype=np.float32)
    f = np.ascontiguousarray(f.transpose(0, 3, 1, 2))                       # T C H W
    T, C, H, W = f.shape
    nh, nw = new_hw
    if (nh, nw) != (H, W):
        y0, y1, wy0, wy1 = _axis(H, nh)
        x0, x1, wx0, wx1 = _axis(W, nw)
        p00, p01 = f[:, :, y0][:, :, :, x0], f[:, :, y0][:, :, :, x1]
        p10, p11 = f[:, :, y1][:, :, :, x0], f[:, :, y1][:, :, :, x1]
        # association of the torch CPU kernel for images of at least ~64x64 outputs (found by bit-comparison with
        # the reference's outputs): rows first, out = fma(top, wy0, bot*wy1), top = fma(p00, wx0, p01*wx1).
        # (Below that size torch takes a four-weight path that differs from this one by <= 2 ulp.)
        top = _fma(p00, wx0, p01 * wx1)
        bot = _fma(p10, wx0, p11 * wx1)
        f = _fma(top, wy0[:, None], bot * wy1[:, None])
    f = f[:, :, y_off:y_off + crop, x_off:x_off + crop]
    if flip:
        f = f[..., ::-1]
    return np.ascontiguousarray(f.transpose(1, 0, 2, 3))


def uniform_crop_offsets(h, w, size, spatial_idx):
    """video_transforms.py:186-201."""
    y = int(math.ceil((h - size) / 2))
    x = int(math.ceil((w - size) / 2))
    if h > w:
        y = 0 if spatial_idx == 0 else (h - size if spatial_idx == 2 else y)
    else:
        x = 0 if spatial_idx == 0 else (w - size if spatial_idx == 2 else x)
    return y, x


# ---- audio: python_speech_features 0.6 ----------------------------------------------------------------------------
def _round_half_up(x):
    return int(decimal.Decimal(x).quantize(decimal.Decimal("1"), rounding=decimal.ROUND_HALF_UP))


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.)


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def filterbank_bins(nfilt, nfft, samplerate, lowfreq=0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    return np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)


def get_filterbanks(nfilt, nfft, samplerate, lowfreq=0, highfreq=None):
    b = filterbank_bins(nfilt, nfft, samplerate, lowfreq, highfreq)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(b[j]), int(b[j + 1])):
            fb[j, i] = (i - b[j]) / (b[j + 1] - b[j])
        for i in range(int(b[j + 1]), int(b[j + 2])):
            fb[j, i] = (b[j + 2] - i) / (b[j + 2] - b[j + 1])
    return fb


def frame_count(slen, frame_len, frame_step):
    return 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))


def logfbank_ref(signal, samplerate, winlen=0.02, winstep=0.01, nfilt=40, nfft=1024, preemph=0.97):
    signal = np.asarray(signal)
    signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])       # sigproc.preemphasis
    frame_len, frame_step = _round_half_up(winlen * samplerate), _round_half_up(winstep * samplerate)
    slen = len(signal)
    nframes = frame_count(slen, frame_len, frame_step)
    padlen = int((nframes - 1) * frame_step + frame_len)
    padded = np.concatenate((signal, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + (np.arange(nframes) * frame_step)[:, None]
    frames = padded[idx]                                                     # rectangular window
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))   # sigproc.powspec
    feat = np.dot(pspec, get_filterbanks(nfilt, nfft, samplerate).T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return np.log(feat)


def get_spec_ref(wav, fr_aud, sample_rate=48000, num_sec=1, aud_spec_type=1, volume=None, z_normalize=False):
    """audio_utils.py:30-72 with the random draws (start sample, volume factor) passed in."""
    to_aud = fr_aud + sample_rate * num_sec
    if fr_aud + (to_aud - fr_aud) > len(wav):
        fr_aud, to_aud = len(wav) - sample_rate * num_sec, len(wav)
    w = wav[fr_aud:to_aud]
    if volume is not None:
        w = w * volume
    spec = logfbank_ref(w, sample_rate, nfilt=40 if aud_spec_type == 1 else 257)
    spec = spec.astype("float32").T[None]
    if z_normalize:
        spec = ((spec - np.float32(1.93)) / np.float32(17.89)).astype(np.float32)
    return spec
