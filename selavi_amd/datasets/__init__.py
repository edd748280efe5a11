"""Device-side input pipeline: the per-clip work of the reference's DataLoader workers
(/root/reference/datasets/video_transforms.py, audio_utils.py) as batched HIP kernels."""
from . import audio_utils, video_transforms  # noqa: F401
