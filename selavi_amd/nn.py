"""Parameter-holder modules with torchvision-compatible names, and the autograd glue.

The module tree reproduces the ``state_dict`` layout of the reference model (torchvision 0.4.2
``r2plus1d_18`` / ``ResNet(BasicBlock)`` as instantiated by /root/reference/model.py:95,114 and the
``MLPv2`` heads of model.py:62-90), so reference checkpoints load by name.  The holders do not
compute: the trunks run through ``selavi_amd.engine`` (one ``autograd.Function`` per trunk), the
heads through ``HeadsFunction`` -- all HIP kernels behind the C ABI.
"""
import math

import os

import torch
from torch import nn

from . import engine, ops
from ._lib import C, ptr, stream


# ------------------------------------------------------------------------------------------ holders
class Conv(nn.Module):
    """Bias-free convolution weight holder (3-D: [Cout,Cin,kt,kh,kw]; 2-D: [Cout,Cin,kh,kw])."""

    def __init__(self, cin, cout, k, stride, pad, dims=3, init="kaiming_fan_out"):
        super().__init__()
        self.in_channels, self.out_channels, self.dims = cin, cout, dims
        if dims == 3:
            self.kernel3, self.stride3, self.padding3 = tuple(k), tuple(stride), tuple(pad)
            shape = (cout, cin) + tuple(k)
        else:
            self.kernel3, self.stride3, self.padding3 = (1,) + tuple(k), (1,) + tuple(stride), (0,) + tuple(pad)
            shape = (cout, cin) + tuple(k)
        self.weight = nn.Parameter(torch.empty(*shape))
        if init == "kaiming_fan_out":      # model.py:54 / torchvision ResNet init
            nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')
        else:                              # default nn.Conv2d init (the replaced audio conv1, model.py:117)
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel={self.kernel3}, stride={self.stride3}, pad={self.padding3}"


class BatchNorm(nn.Module):
    """BatchNorm parameter/buffer holder (gamma=1, beta=0, eps 1e-5, momentum 0.1)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending = 0

    def note_batch(self):
        self._pending += 1          # folded into num_batches_tracked lazily (no per-step kernel)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        self._pending = 0           # the loaded counter already includes whatever was pending when it was saved
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def flush_batches(self):
        """Fold the lazily counted batches into ``num_batches_tracked`` (call before reading the buffer directly)."""
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0
        return self.num_batches_tracked


class BatchNorm3d(BatchNorm):
    pass


class BatchNorm2d(BatchNorm):
    pass


class BatchNorm1d(BatchNorm):
    pass


class ReLU(nn.Module):
    def forward(self, x):            # placeholder keeping torchvision's Sequential indices
        raise RuntimeError("selavi_amd: ReLU is fused into the conv kernels; run the trunk, not its children")


class Identity(nn.Module):           # model.py:43-48
    def forward(self, x):
        return x


class Flatten(nn.Module):            # model.py:25-31
    def forward(self, x):
        return x.view(x.shape[0], -1)


class Unsqueeze(nn.Module):          # model.py:34-40
    def forward(self, x):
        return x.unsqueeze(-1)


def Conv2Plus1D(cin, cout, mid, stride=1):
    return nn.Sequential(
        Conv(cin, mid, (1, 3, 3), (1, stride, stride), (0, 1, 1)), BatchNorm3d(mid), ReLU(),
        Conv(mid, cout, (3, 1, 1), (stride, 1, 1), (1, 0, 0)))


class VideoBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        mid = (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)
        self.conv1 = nn.Sequential(Conv2Plus1D(inplanes, planes, mid, stride), BatchNorm3d(planes), ReLU())
        self.conv2 = nn.Sequential(Conv2Plus1D(planes, planes, mid), BatchNorm3d(planes))
        self.relu = ReLU()
        self.downsample = downsample


class VideoResNet(nn.Module):
    """r2plus1d_18 holder; ``forward`` runs the fused engine (train: batch stats, eval: running)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(
            Conv(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), BatchNorm3d(45), ReLU(),
            Conv(45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)), BatchNorm3d(64), ReLU())
        inpl = 64
        for i, (planes, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)]):
            ds = None
            if stride != 1 or inpl != planes:
                ds = nn.Sequential(Conv(inpl, planes, (1, 1, 1), (stride,) * 3, (0, 0, 0)), BatchNorm3d(planes))
            setattr(self, f"layer{i + 1}", nn.Sequential(VideoBlock(inpl, planes, stride, ds),
                                                          VideoBlock(planes, planes)))
            inpl = planes
        self.avgpool = Identity()
        self.fc = Identity()           # model.py:99
        self.sync = None
        self.precision = "fp32"        # "bf16": the 16-bit MFMA path (ops16; main.py:151 --use_fp16), see AVModel.set_precision

    def forward(self, x):
        u, aux = x.contiguous(), None
        for stage in engine.VIDEO_STAGES:       # one autograd node per stage (see engine.video_stage_forward)
            u = VideoStageFunction.apply(self, stage, u, aux, *_stage_params(self, stage))
            # (the 16-bit stem hands over its RAW output and the scale / shift its consumers apply on load: a second,
            #  non-differentiable output of its node, an input of layer1's)
            u, aux = u if isinstance(u, tuple) else (u, None)
        return u


class AudioBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv(inplanes, planes, (3, 3), (stride, stride), (1, 1), dims=2)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU()
        self.conv2 = Conv(planes, planes, (3, 3), (1, 1), (1, 1), dims=2)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample


class AudioBottleneck(nn.Module):
    """torchvision Bottleneck (1x1 -> 3x3 carrying the stride -> 1x1 to 4 x planes), same attribute / state_dict names
    (model.py:105-106, aud_base_arch = 'resnet50').  The engine sees it as a three-conv chain (engine.audio_forward)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv(inplanes, planes, (1, 1), (1, 1), (0, 0), dims=2)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv(planes, planes, (3, 3), (stride, stride), (1, 1), dims=2)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv(planes, planes * 4, (1, 1), (1, 1), (0, 0), dims=2)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU()
        self.downsample = downsample


class AudioResNet(nn.Module):
    """torchvision ResNet(block, layers) with the 1-channel conv1 of model.py:117-119: BasicBlock for resnet9 / 18 / 34,
    Bottleneck (2048 features) for resnet50."""

    def __init__(self, layers=(1, 1, 1, 1), bottleneck=False):
        super().__init__()
        block, exp = (AudioBottleneck, 4) if bottleneck else (AudioBlock, 1)
        self.feature_dim = 512 * exp
        self.conv1 = Conv(1, 64, (7, 7), (2, 2), (3, 3), dims=2, init="default")
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU()
        self.maxpool = Identity()
        inpl = 64
        for i, (planes, stride, n) in enumerate(zip((64, 128, 256, 512), (1, 2, 2, 2), layers)):
            blocks = []
            for j in range(n):
                s = stride if j == 0 else 1
                ds = None
                if s != 1 or inpl != planes * exp:
                    ds = nn.Sequential(Conv(inpl, planes * exp, (1, 1), (s, s), (0, 0), dims=2), BatchNorm2d(planes * exp))
                blocks.append(block(inpl, planes, s, ds))
                inpl = planes * exp
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.avgpool = Identity()
        self.fc = Identity()
        self.sync = None

    def forward(self, x):
        return TrunkFunction.apply(self, "audio", x.contiguous(), *_trunk_params(self))


def _trunk_params(trunk):
    ps = getattr(trunk, "_plist", None)
    if ps is None:
        ps = trunk._plist = [p for p in trunk.parameters()]
    return ps


def _stage_params(trunk, stage):
    cache = trunk.__dict__.setdefault("_stage_plists", {})
    ps = cache.get(stage)
    if ps is None:
        ps = cache[stage] = [p for p in getattr(trunk, stage).parameters()]
    return ps


def _backend(trunk):
    """Kernel backend of a trunk: fp32 N,C,T,H,W (ops) or bf16 channels-last with fp32 master weights (ops16)."""
    if getattr(trunk, "precision", "fp32") == "bf16":
        from . import ops16
        return ops16
    return ops


def _folded_of(trunk, ectx, need_grad):
    """infer32.FoldedEval attached to this trunk (infer32.folded_eval) -- taken for eval-mode forwards on the fp32 backend only
    (an eval-mode node keeps no autograd state either way: its backward raises, folded or not)."""
    f = getattr(trunk, "_folded_eval", None)
    if f is None or ectx.training or need_grad or ectx.ops is not ops:
        return None
    return f


def _sync_of(mod):
    s = getattr(mod, "sync", None)
    if s == "auto":
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .comm import sync_pair
            return sync_pair(None, getattr(mod, "sync_tag", "bn"))
        return None
    return s


def _take_fork_event(trunk, dfeat):
    """The event a side-stream trunk's backward may fork from: recorded by THIS graph's heads' backward (HeadsFunction stores
    it on the trunk object the model named in its HeadSpec; model.forward clears it), and valid only if the incoming
    gradient IS the heads' dX buffer -- a contiguous view of it, nothing enqueued on the caller's stream after the event
    (a copy made contiguous, an accumulation from a second consumer of the features) produced it."""
    rec = trunk.__dict__.pop("_fork_event", None)
    if rec is None:
        return None
    ev, dptr = rec
    if not dfeat.is_contiguous() or dfeat.data_ptr() != dptr:
        return None
    return ev


def _weight_images(trunk, x, training, first, last_done=False):
    """ops.WeightImages of this trunk for this input shape and mode (fp32 backend only): created at the first stage of a
    pass (``first``: the images are made there, one launch), handed to the later stages of the same pass, built from what the
    first pass recorded once its last stage is through (``last_done``)."""
    from . import ops as _o
    be = _backend(trunk)
    if not _o.BATCH_W_IMAGES or not x.is_cuda:
        return None
    if last_done:
        w = trunk.__dict__.get("_wimg_cur")
        if w is not None and not w.ready:
            w.build(x.device)
        return w
    if first:
        cache = trunk.__dict__.setdefault("_wimg_cache", {})
        key = (tuple(x.shape), bool(training), be.__name__, _o.conv_arithmetic(), _o.benchmark)
        w = cache.get(key)
        if w is None:
            if len(cache) >= 8:
                cache.clear()
            w = cache[key] = be.WeightImages()
        trunk.__dict__["_wimg_cur"] = w
        if w.ready:
            w.run()
        return w
    return trunk.__dict__.get("_wimg_cur")


class TrunkFunction(torch.autograd.Function):
    """One autograd node per trunk: forward = engine schedule, backward = hand-written schedule.

    ``trunk.side_stream`` (AVModel sets it on the audio trunk): the node's kernels run on that HIP stream, forked from
    and joined to the caller's stream with events BY THE NODE ITSELF -- from autograd's point of view the node lives on
    the caller's stream.  (Running ``apply`` under ``torch.cuda.stream(side)`` instead makes autograd hand the incoming
    gradient across streams on its own, which cannot be captured into a HIP graph: hipStreamEndCapture crashes on ROCm
    7.0.)  Forward: forks where it is called and is joined by the caller (model.forward, in front of the heads).
    Backward: forks behind the heads' backward (_take_fork_event) and joins at its own end -- the audio node is the last one
    of the backward pass (it was issued first), so nothing queues behind that join but the optimizer, while its kernels
    run beside the video backward already enqueued."""

    @staticmethod
    def forward(fctx, trunk, kind, x, *params):
        training = trunk.training
        need_grad = training and any(fctx.needs_input_grad)
        ectx = engine.Ctx(training, sync=_sync_of(trunk) if training else None, ops=_backend(trunk))
        ectx.folded = _folded_of(trunk, ectx, need_grad)
        fwd = engine.video_forward if kind == "video" else engine.audio_forward
        side = getattr(trunk, "side_stream", None) if x.is_cuda else None
        if side is not None:
            main = torch.cuda.current_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ectx.wimg = _weight_images(trunk, x, training, first=True)
                feat, saved = fwd(ectx, trunk, x)
                _weight_images(trunk, x, training, first=False, last_done=True)
            x.record_stream(side)
            feat.record_stream(main)
            # the join the consumer of ``feat`` owes: model.forward waits for it in front of the heads (a caller that sets
            # ``side_stream`` itself must do the same; model.forward resets the attribute when it returns)
            trunk._join_event = side.record_event()
        else:
            ectx.wimg = _weight_images(trunk, x, training, first=True)
            feat, saved = fwd(ectx, trunk, x)
            _weight_images(trunk, x, training, first=False, last_done=True)
        fctx.need, fctx.side = need_grad, side
        if need_grad:
            fctx.saved_rec, fctx.trunk, fctx.kind, fctx.sync, fctx.ops = saved, trunk, kind, ectx.sync, ectx.ops
        return feat

    @staticmethod
    def backward(fctx, dfeat):
        if not fctx.need:
            raise RuntimeError("selavi_amd: backward through a trunk that ran in eval / no_grad mode")
        ectx = engine.Ctx(True, sync=fctx.sync, ops=fctx.ops)
        params = _trunk_params(fctx.trunk)
        sink = getattr(fctx.trunk, "grad_sink", None)          # parallel.GradSink: gradients go to its flat buffer
        fresh = False
        if sink is not None:
            # the FIRST call allocates the flat buffer and zero-fills it -- on the caller's stream, now.  A side-stream trunk
            # that forks from the earlier event behind the heads' backward does not wait for that fill: it could land on top
            # of the trunk's first gradients (the dgamma / dbeta of its last BatchNorm: found as a 1-in-10 difference between
            # two two-rank rigs after step 1).  The first step forks from the caller's stream as it is now.
            fresh = (fctx.kind,) not in sink.flat
            ectx.grad_out = sink.views((fctx.kind,), params)
        bwd = engine.video_backward if fctx.kind == "video" else engine.audio_backward
        side = fctx.side
        if side is not None:
            # (no second-level fork: a weight-gradient stream forked from this side stream crashes hipStreamEndCapture on
            #  ROCm 7.0 -- tests/diag/graph_capture_stages.py; the audio trunk's 1.5 ms of small kernels are off the
            #  critical path either way)
            ectx.wgrad_side = False
            main = torch.cuda.current_stream(dfeat.device)
            ev = _take_fork_event(fctx.trunk, dfeat)
            if ev is not None and not fresh and os.environ.get("SELAVI_FORK_EVENT", "1") == "1":
                side.wait_event(ev)                            # behind the heads' backward, not behind the video backward
            else:
                dfeat = dfeat.contiguous()                     # (a copy, if any, is enqueued on main BEFORE the fork)
                side.wait_stream(main)
            with torch.cuda.stream(side):
                bwd(ectx, fctx.saved_rec, dfeat)
            dfeat.record_stream(side)
            for gt in ectx.grads.values():
                gt.record_stream(main)
            main.wait_stream(side)
        else:
            bwd(ectx, fctx.saved_rec, dfeat)
        fctx.saved_rec = None
        if sink is not None:
            sink.deliver((fctx.kind,), params)                 # .grad = bucket views, all-reduce launched
            return (None, None, None) + (None,) * len(params)
        grads = [ectx.grads.get(id(p)) for p in params]
        return (None, None, None) + tuple(grads)


class VideoStageFunction(torch.autograd.Function):
    """One autograd node per stage of the video trunk (stem, layer1..4)."""

    @staticmethod
    def forward(fctx, trunk, stage, x, aux, *params):
        training = trunk.training
        need_grad = training and any(fctx.needs_input_grad)
        ectx = engine.Ctx(training, sync=_sync_of(trunk) if training else None, ops=_backend(trunk))
        ectx.folded = _folded_of(trunk, ectx, need_grad)
        ectx.wimg = _weight_images(trunk, x, training, first=(stage == "stem")) if ectx.folded is None else None
        out, saved = engine.video_stage_forward(ectx, trunk, stage, x, aux)
        if isinstance(out, tuple):
            fctx.mark_non_differentiable(out[1])
        if stage == "layer4":
            _weight_images(trunk, x, training, first=False, last_done=True)
        fctx.need = need_grad
        if need_grad:
            fctx.saved_rec, fctx.trunk, fctx.stage, fctx.sync, fctx.ops = saved, trunk, stage, ectx.sync, ectx.ops
        return out

    @staticmethod
    def backward(fctx, dout, *_daux):
        if not fctx.need:
            raise RuntimeError("selavi_amd: backward through a trunk that ran in eval / no_grad mode")
        ectx = engine.Ctx(True, sync=fctx.sync, ops=fctx.ops)
        params = _stage_params(fctx.trunk, fctx.stage)
        sink = getattr(fctx.trunk, "grad_sink", None)
        if sink is not None:
            ectx.grad_out = sink.views(("video", fctx.stage), params)
        din = engine.video_stage_backward(ectx, fctx.stage, fctx.saved_rec, dout.contiguous())
        fctx.saved_rec = None
        din = din if fctx.needs_input_grad[2] else None
        if sink is not None:
            sink.deliver(("video", fctx.stage), params)
            return (None, None, din, None) + (None,) * len(params)
        grads = [ectx.grads.get(id(p)) for p in params]
        return (None, None, din, None) + tuple(grads)


# ------------------------------------------------------------------------------------------ heads
class MLPv2(nn.Module):
    """model.py:62-90.  ``forward`` (one head, e.g. the SK feature-bank pass of sk_utils.py:309-312)
    runs the MFMA GEMM in eval mode and the grouped kernels (G = 1) in train mode."""

    def __init__(self, n_input, n_classes, n_hidden=512, p=0.3):
        super().__init__()
        self.n_input, self.n_classes, self.n_hidden = n_input, n_classes, n_hidden
        if n_hidden is None:
            self.block_forward = nn.Sequential(Flatten(), nn.Dropout(p=p), nn.Linear(n_input, n_classes, bias=True))
        else:
            self.block_forward = nn.Sequential(
                Flatten(), nn.Dropout(p=p), nn.Linear(n_input, n_hidden, bias=False), Unsqueeze(),
                BatchNorm1d(n_hidden), Flatten(), ReLU(), nn.Dropout(p=p), nn.Linear(n_hidden, n_classes, bias=True))

    def forward(self, x):
        x = x.reshape(x.shape[0], -1).contiguous()
        if self.training:
            out = HeadsFunction.apply(HeadSpec([self], 1, True, self.n_hidden is not None, self.training,
                                               _sync_of(self)), x, x, *head_params([self]))
            return out[0]
        bf = self.block_forward
        if self.n_hidden is None:
            return ops.gemm_nt(x, bf[2].weight, bf[2].bias)
        h = ops.gemm_nt(x, bf[2].weight)
        _, ss = ops.bn_eval_params(bf[4].weight, bf[4].bias, bf[4].running_mean, bf[4].running_var, bf[4].eps)
        C.slv_rowwise_affine(ptr(h), ptr(ss), 1, ptr(h), h.shape[0], h.shape[1], stream())
        return ops.gemm_nt(h, bf[8].weight, bf[8].bias)


class LinearHead(nn.Linear):
    """``nn.Linear(512, K)`` head of the use_mlp=False configuration (model.py:207-208,218-219)."""

    def forward(self, x):
        x = x.reshape(x.shape[0], -1).contiguous()
        return ops.gemm_nt(x, self.weight, self.bias) if not torch.is_grad_enabled() or not self.training else \
            HeadsFunction.apply(HeadSpec([self], 1, True, False, True, None), x, x, *head_params([self]))[0]


def _dropout_stream():
    """(key, counter offset) of the next dropout draw for the library's Philox (slv_dropout_masks): both are drawn from
    torch's default CPU generator, so ``torch.manual_seed`` (opt.py:152 / utils.py:277-283) controls the masks exactly as it
    controls the reference's -- re-seeding replays them -- without a device-side generator state or an ATen launch."""
    v = torch.empty(2, dtype=torch.int64).random_()
    return int(v[0]) & 0xFFFFFFFFFFFFFFFF, int(v[1]) & 0xFFFFFFFFFFFFFFFF


_DROPOUT_DEV_STATE = {}


def dropout_device_state(device, create=False):
    """[seed, offset] of the dropout Philox in DEVICE memory (int64 x 2) for captured steps: a HIP graph freezes the host
    scalars of ``slv_dropout_masks`` into the launch, so every replay would reuse the capture-time masks;
    ``slv_dropout_masks_dev`` reads the pair on the device and advances the offset behind each draw.  Seeded from torch's
    default CPU generator (``torch.manual_seed`` controls it).  Created OUTSIDE the capture (train.GraphedStep)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _DROPOUT_DEV_STATE.get(key)
    if st is None and create:
        seed, off = _dropout_stream()
        st = _DROPOUT_DEV_STATE[key] = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed,
                                                     off - (1 << 64) if off >= (1 << 63) else off],
                                                    dtype=torch.int64, device=dev)
    return st


def _draw_dropout_masks(p, m1, m2, st):
    """Both masks in one launch of the library's own Philox (not torch's generator).  Eager: key / offset are host
    scalars from torch's CPU generator; while the stream is being captured into a graph: the device-resident pair."""
    n2 = m2.numel() if m2 is not None else 0
    if torch.cuda.is_current_stream_capturing():
        state = dropout_device_state(m1.device)
        if state is None:
            raise RuntimeError("dropout under graph capture needs nn.dropout_device_state(device, create=True) before "
                               "the capture (train.GraphedStep does it)")
        C.slv_dropout_masks_dev(ptr(state), float(p), ptr(m1), m1.numel(), ptr(m2), n2, st)
    else:
        C.slv_dropout_masks(*_dropout_stream(), float(p), ptr(m1), m1.numel(), ptr(m2), n2, st)


class HeadSpec:
    def __init__(self, heads, hc, single, has_hidden, training, sync, masks=None, grad_sink=None, fork_for=None):
        self.heads, self.hc, self.single, self.has_hidden = heads, hc, single, has_hidden
        self.training, self.sync, self.masks, self.grad_sink = training, sync, masks, grad_sink
        self.fork_for = fork_for      # the side-stream trunk whose backward may fork behind this node's backward


def _lin_of(head, idx):
    return head.block_forward[idx] if isinstance(head, MLPv2) else head


def head_params(heads):
    ps = []
    for h in heads:
        ps.extend(list(h.parameters()))
    return ps


class HeadList(list):
    """List of per-head logits (the reference returns Python lists, model.py:240-252) that also
    carries the stacked [hc][B][K] tensor so get_loss can run one grouped kernel."""
    stacked = None


class HeadsFunction(torch.autograd.Function):
    """All heads of both modalities in a handful of grouped launches (csrc/heads.hip).

    Inputs: feat_v [B,512], feat_a [B,512], then the parameters of every head in ``spec.heads`` order.
    Output: logits [G][B][K] with G = len(spec.heads); heads[:G/2] read feat_v, heads[G/2:] feat_a
    (``spec.single``: one head reading feat_v)."""

    @staticmethod
    def forward(fctx, spec, feat_v, feat_a, *params):
        heads = spec.heads
        G = len(heads)
        hcg = 1 if spec.single else G // 2          # heads per modality
        B = feat_v.shape[0]
        dev = feat_v.device
        st = stream()
        X = torch.stack([feat_v, feat_a]).contiguous() if not spec.single else feat_v.contiguous().unsqueeze(0)
        shared = 1
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        train = spec.training
        if spec.has_hidden:
            lin1 = [_lin_of(h, 2) for h in heads]
            bns = [h.block_forward[4] for h in heads]
            lin2 = [_lin_of(h, 8) for h in heads]
            p = heads[0].block_forward[1].p if train else 0.0
            IN, HID, K = lin1[0].weight.shape[1], lin1[0].weight.shape[0], lin2[0].weight.shape[0]
            m1 = m2 = None
            msc = 1.0
            if train and p > 0:
                if spec.masks is not None:
                    m1, m2 = spec.masks
                else:
                    m1, m2 = f32(G, B, IN), f32(G, B, HID)
                    _draw_dropout_masks(p, m1, m2, st)
                msc = 1.0 / (1.0 - p)
            W1 = ops.PtrArray([l.weight for l in lin1])
            h = f32(G, B, HID)
            C.slv_heads_linear_fwd(ptr(X), shared, hcg, ptr(m1), msc, W1.p, 0, ptr(h), G, B, IN, HID, st)
            ga, be = ops.PtrArray([b.weight for b in bns]), ops.PtrArray([b.bias for b in bns])
            rm, rv = ops.PtrArray([b.running_mean for b in bns]), ops.PtrArray([b.running_var for b in bns])
            sums = None
            count = float(B)
            if train:
                sums = torch.empty(G, 2, HID, dtype=torch.float64, device=dev)
                C.slv_heads_bn_stats(ptr(h), ptr(sums), G, B, HID, st)
                if spec.sync is not None:
                    with ops._span("syncbn"):
                        ops._allreduce(sums, spec.sync[0])
                    ops.EXCHANGES[0] += 1
                    count *= spec.sync[1]
                for b in bns:
                    b.note_batch()
            a = f32(G, B, HID)
            mi = f32(G, 2, HID)
            C.slv_heads_bn_apply(ptr(h), ptr(sums), count, ga.p, be.p, rm.p, rv.p, ptr(m2), msc, bns[0].momentum,
                                 bns[0].eps, int(train), ptr(a), ptr(mi), G, B, HID, st)
            W2, b2 = ops.PtrArray([l.weight for l in lin2]), ops.PtrArray([l.bias for l in lin2])
            logits = f32(G, B, K)
            C.slv_heads_linear_fwd(ptr(a), 0, hcg, 0, 1.0, W2.p, b2.p, ptr(logits), G, B, HID, K, st)
            fctx.saved = (X, m1, m2, msc, h, a, mi, count)
        else:
            lin = [_lin_of(h, 2) for h in heads]
            p = heads[0].block_forward[1].p if (train and isinstance(heads[0], MLPv2)) else 0.0
            IN, K = lin[0].weight.shape[1], lin[0].weight.shape[0]
            m1, msc = None, 1.0
            if train and p > 0:
                if spec.masks is not None:
                    m1 = spec.masks[0]
                else:
                    m1 = f32(G, B, IN)
                    _draw_dropout_masks(p, m1, None, st)
                msc = 1.0 / (1.0 - p)
            W, bb = ops.PtrArray([l.weight for l in lin]), ops.PtrArray([l.bias for l in lin])
            logits = f32(G, B, K)
            C.slv_heads_linear_fwd(ptr(X), shared, hcg, ptr(m1), msc, W.p, bb.p, ptr(logits), G, B, IN, K, st)
            fctx.saved = (X, m1, None, msc, None, None, None, float(B))
        fctx.spec, fctx.hcg = spec, hcg
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        spec, hcg = fctx.spec, fctx.hcg
        heads = spec.heads
        G = len(heads)
        X, m1, m2, msc, h, a, mi, count = fctx.saved
        fctx.saved = None
        dev = X.device
        st = stream()
        B = X.shape[1]
        dlogits = dlogits.contiguous()
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        grads = {}
        if spec.has_hidden:
            lin1 = [_lin_of(hd, 2) for hd in heads]
            bns = [hd.block_forward[4] for hd in heads]
            lin2 = [_lin_of(hd, 8) for hd in heads]
            IN, HID, K = lin1[0].weight.shape[1], lin1[0].weight.shape[0], lin2[0].weight.shape[0]
            sink = spec.grad_sink
            if sink is not None:          # data parallel: the grouped gradients live in one flat all-reduce buffer
                gflat, (dW2, db2, dga, dbe, dW1) = sink.carve(("heads", G, K, HID, IN), [(G, K, HID), (G, K), (G, HID),
                                                                                          (G, HID), (G, HID, IN)])
            else:
                dW2, db2, dga, dbe, dW1 = f32(G, K, HID), f32(G, K), f32(G, HID), f32(G, HID), f32(G, HID, IN)
            C.slv_heads_linear_bwd_w(ptr(dlogits), ptr(a), 0, hcg, 0, 1.0, ptr(dW2), ptr(db2), G, B, HID, K, st)
            W2 = ops.PtrArray([l.weight for l in lin2])
            da = f32(G, B, HID)
            C.slv_heads_linear_bwd_x(ptr(dlogits), W2.p, 0, 1.0, ptr(da), G, B, HID, K, st)
            ga, be = ops.PtrArray([b.weight for b in bns]), ops.PtrArray([b.bias for b in bns])
            sums = torch.empty(G, 2, HID, dtype=torch.float64, device=dev)
            C.slv_heads_bn_bwd_stats(ptr(da), ptr(h), ptr(mi), ga.p, be.p, ptr(m2), msc, ptr(sums), G, B, HID, st)
            if spec.sync is not None:
                with ops._span("syncbn"):
                    ops._allreduce(sums, spec.sync[0])
                ops.EXCHANGES[0] += 1
            dh = f32(G, B, HID)
            C.slv_heads_bn_bwd_apply(ptr(da), ptr(h), ptr(mi), ga.p, be.p, ptr(m2), msc, ptr(sums), count, ptr(dh),
                                     ptr(dga), ptr(dbe), G, B, HID, st)
            C.slv_heads_linear_bwd_w(ptr(dh), ptr(X), 1, hcg, ptr(m1), msc, ptr(dW1), 0, G, B, IN, HID, st)
            W1 = ops.PtrArray([l.weight for l in lin1])
            dxg = f32(G, B, IN)
            C.slv_heads_linear_bwd_x(ptr(dh), W1.p, ptr(m1), msc, ptr(dxg), G, B, IN, HID, st)
            for g in range(G):
                grads[id(lin1[g].weight)] = dW1[g]
                grads[id(bns[g].weight)] = dga[g]
                grads[id(bns[g].bias)] = dbe[g]
                grads[id(lin2[g].weight)] = dW2[g]
                grads[id(lin2[g].bias)] = db2[g]
        else:
            lin = [_lin_of(hd, 2) for hd in heads]
            IN, K = lin[0].weight.shape[1], lin[0].weight.shape[0]
            sink = spec.grad_sink
            if sink is not None:
                gflat, (dW, db) = sink.carve(("heads", G, K, IN), [(G, K, IN), (G, K)])
            else:
                dW, db = f32(G, K, IN), f32(G, K)
            C.slv_heads_linear_bwd_w(ptr(dlogits), ptr(X), 1, hcg, ptr(m1), msc, ptr(dW), ptr(db), G, B, IN, K, st)
            W = ops.PtrArray([l.weight for l in lin])
            dxg = f32(G, B, IN)
            C.slv_heads_linear_bwd_x(ptr(dlogits), W.p, ptr(m1), msc, ptr(dxg), G, B, IN, K, st)
            for g in range(G):
                grads[id(lin[g].weight)] = dW[g]
                grads[id(lin[g].bias)] = db[g]
        if spec.single:
            dfv, dfa = dxg[0], None
        else:
            dX = f32(2, B, dxg.shape[2])
            C.slv_heads_sum_groups(ptr(dxg), ptr(dX), hcg, B * dxg.shape[2], st)
            dfv, dfa = dX[0], dX[1]
        hp = head_params(heads)
        fork_for = getattr(spec, "fork_for", None)
        if fork_for is not None and dfa is not None:   # where THIS graph's side-stream trunk forks its backward from
            fork_for._fork_event = (torch.cuda.current_stream(dev).record_event(), dfa.data_ptr())
        if spec.grad_sink is not None:        # the grouped gradient tensors are views of one bucket: one all-reduce
            for p in hp:
                p.grad = grads[id(p)]
            spec.grad_sink.reduce([gflat])
            return (None, dfv, dfa) + (None,) * len(hp)
        pg = [grads.get(id(p)) for p in hp]
        return (None, dfv, dfa) + tuple(pg)


class GroupedCE(torch.autograd.Function):
    """mean over heads of mean-over-batch cross entropy (utils.py:377-387) in one kernel."""

    @staticmethod
    def forward(fctx, logits, targets):
        G, B, K = logits.shape
        logits = logits.contiguous()
        tg = targets.reshape(B, -1).contiguous()
        hc = tg.shape[1]
        assert hc == G or hc == 1
        loss_rows = torch.empty(G * B, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        C.slv_heads_ce(ptr(logits), ptr(tg), hc, hc, ptr(loss_rows), ptr(dl), 1.0 / (G * B), G, B, K, stream())
        total = torch.empty((), dtype=torch.float32, device=logits.device)
        C.slv_heads_ce_total(ptr(loss_rows), G * B, 1.0 / (G * B), ptr(total), stream())
        fctx.save_for_backward(dl)
        return total

    @staticmethod
    def backward(fctx, gout):
        (dl,) = fctx.saved_tensors
        return dl * gout, None
