"""clip_augmentation on the GPU (mirrors /root/reference/datasets/video_transforms.py:420-510).

The reference normalises, permutes, resizes (bilinear), crops and flips one clip at a time on the CPU, creating four
intermediate tensors.  Here the random draws are made on the host with the SAME np.random calls in the SAME order
(:52 size, :121-125 crop offsets, :158 flip), and one kernel (csrc/input.hip: slv_clip_augment) reads the uint8
frames once and writes the float32 C x T x S x S clip -- for a whole batch per launch.
"""
import math

import numpy as np
import torch

from .._lib import C, ptr, stream

MEAN = [0.45, 0.45, 0.45]                  # video_transforms.py:13-14
STD = [0.225, 0.225, 0.225]
_MEAN = np.array(MEAN, dtype=np.float32)
_STD = np.array(STD, dtype=np.float32)


def _resized_shape(height, width, size):
    """video_transforms.py:52-67."""
    if (width <= height and width == size) or (height <= width and height == size):
        return height, width
    if width < height:
        return int(math.floor((float(height) / width) * size)), size
    return size, int(math.floor((float(width) / height) * size))


def sample_spatial_params(height, width, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224):
    """The host half of spatial_sampling (:420-459): draws (resized H, resized W, y offset, x offset, flip)
    with the reference's generator calls."""
    assert spatial_idx in [-1, 0, 1, 2, 3, 4, 5]
    size = int(round(np.random.uniform(min_scale, max_scale)))
    nh, nw = _resized_shape(height, width, size)
    if nh < crop_size or nw < crop_size:
        raise ValueError(f"crop {crop_size} does not fit the resized frame {nh}x{nw}")
    if spatial_idx == -1:
        y_off = x_off = 0
        if not (nh == crop_size and nw == crop_size):                       # random_crop :115-125
            if nh > crop_size:
                y_off = int(np.random.randint(0, nh - crop_size))
            if nw > crop_size:
                x_off = int(np.random.randint(0, nw - crop_size))
        flip = bool(np.random.uniform() < 0.5)                               # horizontal_flip :158
        return nh, nw, y_off, x_off, flip
    idx = {0: 0, 1: 1, 2: 2, 3: 0, 4: 1, 5: 2}[spatial_idx]                  # uniform_crop :186-201
    y_off = int(math.ceil((nh - crop_size) / 2))
    x_off = int(math.ceil((nw - crop_size) / 2))
    if nh > nw:
        y_off = 0 if idx == 0 else (nh - crop_size if idx == 2 else y_off)
    else:
        x_off = 0 if idx == 0 else (nw - crop_size if idx == 2 else x_off)
    flip = spatial_idx in (3, 4, 5)
    if flip:
        np.random.uniform()                                                  # horizontal_flip(1, ...) still draws
    return nh, nw, y_off, x_off, flip


def clip_augmentation_batch(clips, params, crop_size, out=None):
    """clips: list of uint8 device tensors T x H x W x 3 (same T; H, W may differ per clip), or one B x T x H x W x 3
    tensor.  params: per clip (resized H, resized W, y offset, x offset, flip).  -> B x 3 x T x S x S float32."""
    if torch.is_tensor(clips):
        assert clips.dtype == torch.uint8 and clips.dim() == 5 and clips.shape[-1] == 3 and clips.is_cuda
        B, T, H, W = clips.shape[:4]
        buf = clips.contiguous()
        shapes = [(H, W)] * B
        offs = [b * T * H * W * 3 for b in range(B)]
    else:
        B = len(clips)
        T = clips[0].shape[0]
        shapes, offs, o = [], [], 0
        for c in clips:
            assert c.dtype == torch.uint8 and c.dim() == 4 and c.shape[-1] == 3 and c.shape[0] == T and c.is_cuda
            shapes.append((c.shape[1], c.shape[2]))
            offs.append(o)
            o += c.numel()
        buf = torch.cat([c.reshape(-1) for c in clips])
    desc = np.zeros((B, 8), dtype=np.int64)
    for b, ((H, W), (nh, nw, yo, xo, flip)) in enumerate(zip(shapes, params)):
        if not (0 <= yo and yo + crop_size <= nh and 0 <= xo and xo + crop_size <= nw):
            raise ValueError("crop window outside the resized frame")
        desc[b] = (offs[b], H, W, nh, nw, yo, xo, int(flip))
    desc_d = torch.from_numpy(desc).to(buf.device, non_blocking=False)
    if out is None:
        out = torch.empty((B, 3, T, crop_size, crop_size), dtype=torch.float32, device=buf.device)
    assert out.shape == (B, 3, T, crop_size, crop_size) and out.dtype == torch.float32
    C.slv_clip_augment(ptr(buf), ptr(desc_d), ptr(out), B, T, crop_size, _MEAN.ctypes.data, _STD.ctypes.data, stream())
    return out


def clip_augmentation(frames, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224, colorjitter=False,
                      use_grayscale=False, use_gaussian=False):
    """One clip, the reference's signature (:462-471): frames T x H x W x 3 uint8 (device) -> 3 x T x S x S float32."""
    if colorjitter or use_grayscale or use_gaussian:
        raise NotImplementedError("colour jitter / grayscale / gaussian are off in the reference's defaults "
                                  "(opt.py:47-52) and are not part of the device pipeline")
    prm = sample_spatial_params(frames.shape[1], frames.shape[2], spatial_idx, min_scale, max_scale, crop_size)
    return clip_augmentation_batch([frames], [prm], crop_size)[0]
