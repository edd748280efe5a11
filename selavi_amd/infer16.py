"""Eval-mode trunk forward in bf16 (experimental, opt-in): the Sinkhorn-Knopp feature pass.

Under the reference's schedule the eval forward over the whole dataset (sk_utils.py:137-233) is a quarter of the wall
clock and at fp32 it is MFMA-bound (bench ``sk_round``).  In eval mode every BatchNorm is a fixed per-channel affine,
so each conv + BN (+ residual) + ReLU of torchvision's blocks is ONE launch of csrc/conv_cl16.hip on bf16
channels-last activations; nothing else runs between the input conversion and the average pool.

    eng = infer16.Engine(model)            # re-lays-out the weights (repeat after the weights change)
    feat_v, feat_a = eng.features(video, audio)      # fp32 [B, 512] each, what return_features=True returns

Numerics: bf16 activations and weights, fp32 accumulation and epilogue -- NOT the bit-exact path: pseudo labels
computed from these features can differ from the fp32 ones for samples near a decision boundary
(tests/test_infer16_gpu.py quantifies it).  Off by default; SELAVI_FEATURE_PASS=bf16 switches the SK round to it.
"""
import torch

from . import ops, ops16
from ._lib import C, ptr, stream


def _affine(bn):
    _, ss = ops.bn_eval_params(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    return ss                                    # [2][C]: scale, shift


class _Layer:
    def __init__(self, conv, bn):
        self.conv = ops16.Conv16(conv.weight.detach(), conv.stride3, conv.padding3)
        self.ss = _affine(bn)

    def __call__(self, x, res=None, relu=True):
        return self.conv(x, scale_shift=self.ss, res=res, relu=relu)


class _Stem:
    def __init__(self, conv, bn):
        self.conv = ops16.StemConv16(conv.weight.detach(), conv.stride3, conv.padding3)
        self.ss = _affine(bn)

    def __call__(self, x):
        return self.conv(x, scale_shift=self.ss, relu=True)


class Engine:
    def __init__(self, model):
        m = model.module if hasattr(model, "module") else model
        v, a = m.video_network.base, m.audio_network.base
        with torch.no_grad():
            self.v_stem = [_Stem(v.stem[0], v.stem[1]), _Layer(v.stem[3], v.stem[4])]
            self.v_blocks = []
            for li in range(1, 5):
                for blk in getattr(v, f"layer{li}"):
                    c1, c2 = blk.conv1, blk.conv2          # Sequential(Conv2Plus1D, BN, ReLU) / Sequential(Conv2Plus1D, BN)
                    chain = [_Layer(c1[0][0], c1[0][1]), _Layer(c1[0][3], c1[1]),
                             _Layer(c2[0][0], c2[0][1]), _Layer(c2[0][3], c2[1])]
                    ds = _Layer(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                    self.v_blocks.append((chain, ds))
            self.a_stem = _Stem(a.conv1, a.bn1)
            self.a_blocks = []
            for li in range(1, 5):
                for blk in getattr(a, f"layer{li}"):
                    ds = _Layer(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                    self.a_blocks.append(([_Layer(blk.conv1, blk.bn1), _Layer(blk.conv2, blk.bn2)], ds))

    @staticmethod
    def _pool(x, channels):
        N, T, H, W, Cp = x.shape
        out = torch.empty((N, channels), dtype=torch.float32, device=x.device)
        C.slv_avgpool_cl16(ptr(x), ptr(out), N, T * H * W, channels, Cp, stream())
        return out

    @torch.no_grad()
    def video_features(self, video):
        x = self.v_stem[1](self.v_stem[0](video))               # the stem's first conv converts the fp32 clip itself
        for chain, ds in self.v_blocks:
            y = x
            for l in chain[:-1]:
                y = l(y)
            shortcut = x if ds is None else ds(x, relu=False)
            x = chain[-1](y, res=shortcut, relu=True)          # relu(bn2(conv) + shortcut)
        return self._pool(x, 512)

    @torch.no_grad()
    def audio_features(self, spec):
        x = self.a_stem(spec)
        N, _, H, W, Cp = x.shape
        y = torch.empty((N, 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cp), dtype=torch.bfloat16, device=x.device)
        C.slv_maxpool_cl16(ptr(x), ptr(y), N, H, W, Cp, stream())
        x = y
        for chain, ds in self.a_blocks:
            shortcut = x if ds is None else ds(x, relu=False)
            x = chain[1](chain[0](x), res=shortcut, relu=True)
        return self._pool(x, 512)

    def features(self, video, audio):
        return self.video_features(video), self.audio_features(audio)
