"""AVModel facade -- drop-in for /root/reference/model.py (``load_model`` / ``AVModel.forward``).

Same constructor arguments, attributes (``return_features``, ``use_mlp``, ``hc``, ``mlp_v{h}``,
``mlp_a{h}``, ``video_network.base.{stem,layer1..4}``, ``audio_network.base``) and ``state_dict``
keys as the reference, so ``main.py``'s loop (main.py:105-114,284) and ``sk_utils`` (:187,:266-282)
run unchanged; the arithmetic is libselavi_hip.so.
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

from . import nn as snn


def get_video_feature_extractor(vid_base_arch='r2plus1d_18', pretrained=False, duration=1):   # model.py:93-100
    assert vid_base_arch == 'r2plus1d_18', "only r2plus1d_18 is on the hot path"
    assert not pretrained, "no pretrained weights in this build (no network)"
    return snn.VideoResNet()


def get_audio_feature_extractor(aud_base_arch='resnet9', pretrained=False, duration=1):       # model.py:103-121
    layers = {'resnet9': (1, 1, 1, 1), 'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet50': (3, 4, 6, 3)}
    assert aud_base_arch in layers
    return snn.AudioResNet(layers[aud_base_arch], bottleneck=aud_base_arch == 'resnet50')


class VideoBaseNetwork(nn.Module):    # model.py:135-149
    def __init__(self, vid_base_arch='r2plus1d_18', pretrained=False, norm_feat=False, duration=1):
        super().__init__()
        self.base = get_video_feature_extractor(vid_base_arch, pretrained, duration)
        self.norm_feat = norm_feat

    def forward(self, x):
        x = self.base(x).squeeze()
        return F.normalize(x, p=2, dim=1) if self.norm_feat else x


class AudioBaseNetwork(nn.Module):    # model.py:152-166
    def __init__(self, aud_base_arch='resnet9', pretrained=False, norm_feat=False, duration=1):
        super().__init__()
        self.base = get_audio_feature_extractor(aud_base_arch, pretrained, duration)
        self.norm_feat = norm_feat

    def forward(self, x):
        x = self.base(x).squeeze()
        return F.normalize(x, p=2, dim=1) if self.norm_feat else x


_SIDE_STREAMS = {}      # device -> HIP stream of the audio trunk


class AVModel(nn.Module):             # model.py:169-252
    def __init__(self, vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', pretrained=False, norm_feat=True,
                 use_mlp=False, headcount=1, num_classes=256, use_max_pool=False):
        super().__init__()
        self.use_mlp, self.hc, self.norm_feat = use_mlp, headcount, norm_feat
        self.return_features = False
        self.video_network = VideoBaseNetwork(vid_base_arch, pretrained=pretrained)
        self.audio_network = AudioBaseNetwork(aud_base_arch, pretrained=pretrained)
        mk = (lambda: snn.MLPv2(512, num_classes, n_hidden=512)) if use_mlp else \
            (lambda: snn.LinearHead(512, num_classes))
        if self.hc == 1:
            self.mlp_v, self.mlp_a = mk(), mk()
        else:
            for a in range(self.hc):
                setattr(self, "mlp_v%d" % a, mk())
                setattr(self, "mlp_a%d" % a, mk())
        self._dropout_masks = None      # tests may inject (m1, m2) [G][B][512] float masks
        self.overlap_audio = os.environ.get("SELAVI_OVERLAP_AUDIO", "1") == "1"
        self.set_sync_bn("auto")

    # ---- SyncBN (main.py:117-118 converts every BN; here it is a switch on the fused BN kernels)
    def set_sync_bn(self, mode, group=None):
        """mode: 'auto' (sync iff torch.distributed is initialised with world_size > 1), True, False."""
        self.video_network.base.sync_tag, self.audio_network.base.sync_tag = "bn", "bn_audio"
        sync_audio = None
        if mode is True:
            from .comm import sync_pair
            # RCCL behind the C ABI when the backend is nccl, else the torch group.  The audio trunk issues its exchanges
            # from its own stream next to the video trunk's: with the native path it gets a communicator of its own
            # (collectives on one RCCL communicator serialise in issue order, whatever stream they are on)
            sync_audio = sync_pair(group, "bn_audio" if self.overlap_audio else "bn")
            sync = sync_pair(group, "bn")
        elif mode == "auto":
            sync = sync_audio = "auto"
        else:
            sync = None
        self._sync = sync
        self.video_network.base.sync = sync
        self.audio_network.base.sync = sync_audio
        for h in self._heads():
            h.sync = sync

    def set_precision(self, precision, audio=None):
        """"fp32" (default; every parity claim) or "bf16": both trunks train on the 16-bit MFMA path (what --use_fp16 / apex O1
        does to the convs in the reference, main.py:151-153): bf16 channels-last activations and activation gradients, fp32
        accumulation, fp32 master weights, BatchNorm statistics and parameters in fp32, no loss scaling (bf16 has fp32's
        exponent range).  ``audio``: precision of the ResNet-9 audio trunk when it should differ from the video trunk's
        (its 2-D convs are T = 1 launches of the same kernels; SELAVI_AUDIO_PRECISION overrides the default).  The heads
        (0.2 % of the FLOPs) stay in fp32, on the matrix cores."""
        assert precision in ("fp32", "bf16")
        if audio is None:
            audio = os.environ.get("SELAVI_AUDIO_PRECISION", precision)
        assert audio in ("fp32", "bf16")
        self.video_network.base.precision = precision
        self.audio_network.base.precision = audio

    def set_grad_sink(self, sink):
        """parallel.DataParallel: parameter gradients are written into the sink's flat buffers and all-reduced per
        autograd node (None: gradients are returned to autograd as usual)."""
        self._grad_sink = sink
        self.video_network.base.grad_sink = sink
        self.audio_network.base.grad_sink = sink

    @staticmethod
    def _side_stream(device):
        st = _SIDE_STREAMS.get(device)
        if st is None:
            st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
        return st

    def _heads(self):
        if self.hc == 1:
            return [self.mlp_v, self.mlp_a]
        return [getattr(self, "mlp_v%d" % h) for h in range(self.hc)] + \
               [getattr(self, "mlp_a%d" % h) for h in range(self.hc)]

    def forward(self, img, spec, whichhead=0):
        # The audio trunk (0.6 % of the FLOPs, ~70 small launches that cannot fill 256 CUs) runs on its own
        # HIP stream next to the video trunk, forward and -- autograd replays a node on the stream of its
        # forward -- backward.  It is issued first: autograd then runs the video backward before the
        # audio backward, which leaves the tail of the video gradients' all-reduce something to hide behind.
        atrunk = self.audio_network.base
        atrunk.__dict__.pop("_fork_event", None)                  # a fork point belongs to ONE graph (nn._take_fork_event)
        overlapped = False
        if self.overlap_audio and spec.is_cuda:
            main = torch.cuda.current_stream(spec.device)
            # the attribute lives for THIS call only: a later direct ``model.audio_network(spec)`` (feature extraction,
            # tools) must run on the caller's stream, where nobody would join a side stream for it
            atrunk.side_stream = self._side_stream(spec.device)
            atrunk._join_event = None
            try:
                aud_features = self.audio_network(spec).squeeze()
                img_features = self.video_network(img).squeeze()
            finally:
                atrunk.side_stream = None
                ev, atrunk._join_event = atrunk._join_event, None
                if ev is not None:
                    main.wait_event(ev)                           # the join of the node's forward fork
                    overlapped = True
        else:
            atrunk.side_stream = None
            aud_features = self.audio_network(spec).squeeze()
            img_features = self.video_network(img).squeeze()
        if self.return_features:                                  # model.py:226-227
            return img_features, aud_features
        if aud_features.shape[-1] != 512:
            # the reference builds its heads for encoder_dim_a = 512 (model.py:198-199): its nn.Linear raises the same way
            raise RuntimeError("size mismatch: the heads take 512-d audio features, this audio trunk yields %d "
                               "(use return_features)" % aud_features.shape[-1])
        if aud_features.dim() == 1:
            aud_features = aud_features.unsqueeze(0)
        if img_features.dim() == 1:
            img_features = img_features.unsqueeze(0)
        heads = self._heads()
        spec_ = snn.HeadSpec(heads, self.hc, False, self.use_mlp, self.training,
                             snn._sync_of(self.video_network.base) if self.training else None,
                             masks=self._dropout_masks, grad_sink=getattr(self, "_grad_sink", None),
                             fork_for=atrunk if overlapped else None)
        logits = snn.HeadsFunction.apply(spec_, img_features.contiguous(), aud_features.contiguous(),
                                         *snn.head_params(heads))
        if self.norm_feat:
            logits = F.normalize(logits, p=2, dim=2)
        if self.hc == 1:
            return logits[0], logits[1]
        outs1, outs2 = snn.HeadList(logits[h] for h in range(self.hc)), \
            snn.HeadList(logits[self.hc + h] for h in range(self.hc))
        outs1.stacked, outs2.stacked = logits[:self.hc], logits[self.hc:]
        return outs1, outs2


def load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', pretrained=False, norm_feat=True,
               use_mlp=False, headcount=1, num_classes=256, use_max_pool=False):   # model.py:255-275
    return AVModel(vid_base_arch=vid_base_arch, aud_base_arch=aud_base_arch, pretrained=pretrained,
                   norm_feat=norm_feat, use_mlp=use_mlp, headcount=headcount, num_classes=num_classes,
                   use_max_pool=use_max_pool)
