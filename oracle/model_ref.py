"""Pure-torch CPU restatement of the SeLaVi networks (TEST INFRASTRUCTURE ONLY).

What is restated and from where
-------------------------------
* ``r2plus1d_18`` / ``resnet`` trunks: torchvision==0.4.2 (``environment.yml:166``), called from
  ``/root/reference/model.py:95`` (``torchvision.models.video.__dict__['r2plus1d_18']``) and
  ``model.py:106,114-115`` (``torchvision.models.resnet._resnet(arch, BasicBlock, layers, ...)``).
  torchvision is not vendored in the reference and not installed in the build image, so the
  published architecture is restated here with *the same module
  names*, which makes ``state_dict()`` keys identical to the reference checkpoints.
* ``MLPv2`` head, ``AVModel`` forward, ``load_model``: ``model.py:62-90,169-252,255-275``.
* ``get_loss``: ``utils.py:377-387``.

Every module here is a plain ``torch.nn`` module executed by torch's CPU kernels; this is
the arithmetic oracle the HIP kernels are compared against (fp32, tolerance 1e-3).
"""
import math

import torch
from torch import nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# R(2+1)D-18 (torchvision.models.video.resnet, restated)
# --------------------------------------------------------------------------------------
class Conv2Plus1D(nn.Sequential):
    """(1,3,3) spatial conv -> BN -> ReLU -> (3,1,1) temporal conv, all bias-free."""

    def __init__(self, in_planes, out_planes, midplanes, stride=1, padding=1):
        super().__init__(
            nn.Conv3d(in_planes, midplanes, kernel_size=(1, 3, 3), stride=(1, stride, stride),
                      padding=(0, padding, padding), bias=False),
            nn.BatchNorm3d(midplanes),
            nn.ReLU(inplace=True),
            nn.Conv3d(midplanes, out_planes, kernel_size=(3, 1, 1), stride=(stride, 1, 1),
                      padding=(padding, 0, 0), bias=False),
        )


class VideoBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # one midplanes value per block, shared by conv1 and conv2
        midplanes = (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)
        self.conv1 = nn.Sequential(
            Conv2Plus1D(inplanes, planes, midplanes, stride), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(
            Conv2Plus1D(planes, planes, midplanes), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.conv1(x)
        out = self.conv2(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


class R2Plus1dStem(nn.Sequential):
    def __init__(self):
        super().__init__(
            nn.Conv3d(3, 45, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False),
            nn.BatchNorm3d(45),
            nn.ReLU(inplace=True),
            nn.Conv3d(45, 64, kernel_size=(3, 1, 1), stride=(1, 1, 1), padding=(1, 0, 0), bias=False),
            nn.BatchNorm3d(64),
            nn.ReLU(inplace=True),
        )


class VideoResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_classes=400):
        super().__init__()
        self.inplanes = 64
        self.stem = R2Plus1dStem()
        self.layer1 = self._make_layer(64, layers[0], stride=1)
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(512, num_classes)
        self._initialize_weights()

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv3d(self.inplanes, planes, kernel_size=1, stride=(stride, stride, stride), bias=False),
                nn.BatchNorm3d(planes))
        layers = [VideoBasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(VideoBasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.stem(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        x = self.avgpool(x)
        x = x.flatten(1)
        return self.fc(x)


def r2plus1d_18(pretrained=False, progress=True, **kwargs):
    assert not pretrained, "no network in the build image"
    return VideoResNet(**kwargs)


# --------------------------------------------------------------------------------------
# 2-D ResNet (torchvision.models.resnet, restated: BasicBlock + ResNet + _resnet)
# --------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (the "v1.5" form torchvision ships: the stride sits on the 3x3 conv), groups = 1,
    base_width = 64 -> width = planes.  Used by model.py:105-106 for aud_base_arch = 'resnet50'."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=None):
        super().__init__()
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        x = torch.flatten(x, 1)
        return self.fc(x)


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    """torchvision 0.4.x private factory with the 5-positional-arg signature model.py:114 uses."""
    assert not pretrained
    return ResNet(block, layers, **kwargs)


def resnet18(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet18', BasicBlock, [2, 2, 2, 2], pretrained, progress, **kwargs)


def resnet34(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet34', BasicBlock, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet50(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


# --------------------------------------------------------------------------------------
# model.py restated: heads, AVModel, load_model
# --------------------------------------------------------------------------------------
class Flatten(nn.Module):          # model.py:25-31
    def forward(self, x):
        return x.view(x.shape[0], -1)


class Unsqueeze(nn.Module):        # model.py:34-40
    def forward(self, x):
        return x.unsqueeze(-1)


class Identity(nn.Module):         # model.py:43-48
    def forward(self, x):
        return x


def random_weight_init(model):     # model.py:51-59
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out')
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm3d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class MLPv2(nn.Module):            # model.py:62-90
    def __init__(self, n_input, n_classes, n_hidden=512, p=0.3):
        super().__init__()
        self.n_input, self.n_classes, self.n_hidden = n_input, n_classes, n_hidden
        if n_hidden is None:
            self.block_forward = nn.Sequential(Flatten(), nn.Dropout(p=p), nn.Linear(n_input, n_classes, bias=True))
        else:
            self.block_forward = nn.Sequential(
                Flatten(), nn.Dropout(p=p), nn.Linear(n_input, n_hidden, bias=False), Unsqueeze(),
                nn.BatchNorm1d(n_hidden), Flatten(), nn.ReLU(inplace=True), nn.Dropout(p=p),
                nn.Linear(n_hidden, n_classes, bias=True))

    def forward(self, x):
        return self.block_forward(x)


def get_video_feature_extractor(vid_base_arch='r2plus1d_18', pretrained=False):   # model.py:93-100
    assert vid_base_arch == 'r2plus1d_18'
    model = r2plus1d_18(pretrained=pretrained)
    if not pretrained:
        random_weight_init(model)
    model.fc = Identity()
    return model


def get_audio_feature_extractor(aud_base_arch='resnet9'):                         # model.py:103-121
    assert aud_base_arch in ('resnet9', 'resnet18', 'resnet34', 'resnet50')
    if aud_base_arch == 'resnet9':                                                # model.py:112-115
        model = _resnet(aud_base_arch, BasicBlock, [1, 1, 1, 1], False, False)
    else:                                                                         # model.py:105-106
        model = {'resnet18': resnet18, 'resnet34': resnet34, 'resnet50': resnet50}[aud_base_arch](pretrained=False)
    # conv1 is replaced AFTER the kaiming init -> default nn.Conv2d init for this layer
    model.conv1 = nn.Conv2d(1, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3), bias=False)
    model.fc = Identity()
    return model


class VideoBaseNetwork(nn.Module):   # model.py:135-149
    def __init__(self, vid_base_arch='r2plus1d_18', pretrained=False, norm_feat=False):
        super().__init__()
        self.base = get_video_feature_extractor(vid_base_arch, pretrained)
        self.norm_feat = norm_feat

    def forward(self, x):
        x = self.base(x).squeeze()
        return F.normalize(x, p=2, dim=1) if self.norm_feat else x


class AudioBaseNetwork(nn.Module):   # model.py:152-166
    def __init__(self, aud_base_arch='resnet9', pretrained=False, norm_feat=False):
        super().__init__()
        self.base = get_audio_feature_extractor(aud_base_arch)
        self.norm_feat = norm_feat

    def forward(self, x):
        x = self.base(x).squeeze()
        return F.normalize(x, p=2, dim=1) if self.norm_feat else x


class AVModel(nn.Module):            # model.py:169-252
    def __init__(self, vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', pretrained=False,
                 norm_feat=True, use_mlp=False, headcount=1, num_classes=256, use_max_pool=False):
        super().__init__()
        self.use_mlp, self.hc, self.norm_feat = use_mlp, headcount, norm_feat
        self.return_features = False
        self.video_network = VideoBaseNetwork(vid_base_arch, pretrained)
        self.audio_network = AudioBaseNetwork(aud_base_arch, pretrained)

        def head():
            # model.py:204-205,214-215 -- BOTH modalities get MLPv2(512, K, n_hidden=512) when use_mlp
            return MLPv2(512, num_classes, n_hidden=512) if use_mlp else nn.Linear(512, num_classes)
        if self.hc == 1:
            self.mlp_v = head()
            self.mlp_a = MLPv2(512, num_classes) if use_mlp else nn.Linear(512, num_classes)
        else:
            for a in range(self.hc):
                setattr(self, "mlp_v%d" % a, head())
                setattr(self, "mlp_a%d" % a, MLPv2(512, num_classes) if use_mlp else nn.Linear(512, num_classes))

    def forward(self, img, spec, whichhead=0):
        img_features = self.video_network(img).squeeze()
        aud_features = self.audio_network(spec).squeeze()
        if self.return_features:
            return img_features, aud_features
        if aud_features.dim() == 1:
            aud_features = aud_features.unsqueeze(0)
        if img_features.dim() == 1:
            img_features = img_features.unsqueeze(0)
        if self.hc == 1:
            v, a = self.mlp_v(img_features), self.mlp_a(aud_features)
            if self.norm_feat:
                v, a = F.normalize(v, p=2, dim=1), F.normalize(a, p=2, dim=1)
            return v, a
        outs1, outs2 = [], []
        for h in range(self.hc):
            v = getattr(self, "mlp_v%d" % h)(img_features)
            a = getattr(self, "mlp_a%d" % h)(aud_features)
            if self.norm_feat:
                v, a = F.normalize(v, p=2, dim=1), F.normalize(a, p=2, dim=1)
            outs1.append(v)
            outs2.append(a)
        return outs1, outs2


def load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', pretrained=False, norm_feat=True,
               use_mlp=False, headcount=1, num_classes=256, use_max_pool=False):   # model.py:255-275
    return AVModel(vid_base_arch, aud_base_arch, pretrained, norm_feat, use_mlp, headcount,
                   num_classes, use_max_pool)


def get_loss(activations, targets, headcount=1):   # utils.py:377-387
    if headcount == 1:
        return F.cross_entropy(activations, targets)
    return torch.mean(torch.stack(
        [F.cross_entropy(activations[h], targets[:, h]) for h in range(headcount)]))


# --------------------------------------------------------------------------------------
# portable deterministic parameter / input generator (shared by oracle, tests and bench)
# --------------------------------------------------------------------------------------
def portable_fill_(t, seed, scale=1.0, kind="normal"):
    """Fill ``t`` in place from a counter-based generator that is identical on every host.

    torch's CPU and GPU RNG streams differ, and fixtures must be regenerable on the GPU
    box without /root/reference, so weights/inputs of end-to-end fixtures come from this
    splitmix64-style hash of (seed, flat index) -> Box-Muller normal / uniform(-1, 1).
    """
    import numpy as np
    n = t.numel()
    idx = np.arange(n, dtype=np.uint64)

    def mix(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over='ignore'):
        a = mix(idx * np.uint64(2) + np.uint64(seed) * np.uint64(0x100000001B3))
        b = mix(idx * np.uint64(2) + np.uint64(1) + np.uint64(seed) * np.uint64(0x100000001B3))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    if kind == "normal":
        v = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    else:
        v = 2.0 * u1 - 1.0
    v = (v * scale).astype(np.float32 if t.dtype == torch.float32 else np.float64)
    with torch.no_grad():
        t.copy_(torch.from_numpy(v).view(t.shape))
    return t


def portable_init_(model, seed=31):
    """Deterministic re-initialisation of every parameter of an AVModel-shaped module.

    Keeps the reference's init *family* per layer (kaiming fan_out normal for convs,
    uniform(+-1/sqrt(fan_in)) for Linear and the replaced audio conv1, BN gamma=1 beta=0 with a
    small perturbation so BN-affine bugs are visible) but draws from ``portable_fill_``.
    Iterates ``named_parameters`` in sorted-name order so oracle and HIP facade agree.
    """
    import zlib
    for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
        s = (zlib.crc32(name.encode()) + seed * 7919) & 0x7FFFFFFF
        if p.dim() >= 4:                                   # conv weights
            fan_out = p.shape[0] * int(torch.tensor(p.shape[2:]).prod())
            if name.endswith("audio_network.base.conv1.weight"):
                fan_in = p.shape[1] * int(torch.tensor(p.shape[2:]).prod())
                portable_fill_(p.data, s, scale=1.0 / math.sqrt(fan_in), kind="uniform")
            else:
                portable_fill_(p.data, s, scale=math.sqrt(2.0 / fan_out), kind="normal")
        elif p.dim() == 2:                                 # Linear weights
            portable_fill_(p.data, s, scale=1.0 / math.sqrt(p.shape[1]), kind="uniform")
        elif name.endswith("weight"):                      # BN gamma
            portable_fill_(p.data, s, scale=0.1, kind="uniform")
            p.data.add_(1.0)
        else:                                              # BN beta / Linear bias
            portable_fill_(p.data, s, scale=0.1, kind="uniform")
    return model
