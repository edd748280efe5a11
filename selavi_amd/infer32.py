"""Eval-mode trunk forward in fp32 with BatchNorm FOLDED into the conv weights: the Sinkhorn-Knopp feature pass.

The reference runs the model in eval mode over the whole dataset before every pseudo-label round
(/root/reference/src/sk_utils.py:137-254); under its schedule that pass is a fifth of the wall clock (bench ``sk_round``).
In eval mode a BatchNorm is a fixed per-channel affine, and the weights do not change during the pass, so

    relu(bn(conv(x, w)))          = relu(conv(x, w * s) + h)             s = gamma / sqrt(var + eps), h = beta - mean * s
    relu(bn(conv(x, w)) + shortcut) = relu(conv(x, w * s) + h + shortcut)

is ONE launch of the split-operand implicit-GEMM kernel with its EPI_EVAL epilogue (csrc/igemm3.hpp, slv_conv_fwd_eval):
no BatchNorm + ReLU prologue on the consumer's loads, no statistics, no block-tail pass over (y, shortcut, out), and the
weight images (w * s cut into bf16 pieces) are made ONCE per pass instead of once per batch.  Activations stay the
reference's fp32 N,C,T,H,W tensors.  The two 3 / 1-channel stem convs have no split-operand image: they keep the native
fp32 kernel followed by slv_bn_act.

Both forms are OPT-IN for the SK round (SELAVI_FEATURE_PASS / args.feature_pass); its default stays the model's own eval forward.
``pieces``: 3 ("fp32_folded") = the exact three-piece operand split of the training path (six partial products per fp32
product: the arithmetic every parity claim of the fp32 path rests on) -- features 6e-7 off the plain forward and, measured,
no faster (2 131 against 2 120 clips/s at 64 clips: the conv main loops are the time, not the passes folded away); 2 = two
pieces / three partial products ("fp32x2": 2 800 clips/s): 16-17 significand bits per product at half the matrix-core work -- features move by ~2e-4
relative (41 convs deep), inside the 1e-3 the north star allows for logits, but pseudo labels are an argmax and are only
guaranteed identical with pieces = 3 (tests/test_infer32_gpu.py quantifies both).

    with infer32.folded_eval(model, pieces=3):       # model in eval mode, under torch.no_grad()
        feat_v, feat_a = model(video, audio)         # every eval forward inside the block takes the folded path

The images are cached on the context: leave the block before the weights change.
"""
import contextlib

import torch

from . import ops
from ._lib import C, ptr, stream


def _round16_(w):
    """fp32 -> the nearest value with 16 significand bits (round to nearest even), in place.  Such a value is EXACTLY the sum of
    the first two pieces of the three-piece split (the third is zero), so the two-piece kernel reads its weights without the
    one-sided truncation error of dropping a non-zero third piece: 40 layers of a -2^-17 relative bias added up coherently to
    3.5e-4 on the features; rounded, the weights' error is unbiased (measured: tests/test_infer32_gpu.py)."""
    b = w.view(torch.int32)
    b.add_(0x7F + ((b >> 8) & 1)).bitwise_and_(~0xFF)
    return w


class _Folded:
    """One conv + BatchNorm pair: scaled weights, bias, and the split-operand image (made at first use)."""
    __slots__ = ("w", "bias", "ss", "img", "ok", "cfg")

    def __init__(self, conv, bn, pieces):
        _, ss = ops.bn_eval_params(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        w = conv.weight.detach()
        self.ss = ss
        self.bias = ss[1].contiguous()
        self.w = (w * ss[0].view(-1, *([1] * (w.dim() - 1)))).contiguous()       # one-time weight preparation
        if pieces == 2:
            _round16_(self.w)
        self.img = None
        self.ok = None
        self.cfg = {}               # plan -> launch configuration of the eval kernel (benchmark mode: timed at first use)


class FoldedEval:
    def __init__(self, pieces=3):
        assert pieces in (2, 3)
        self.pieces = pieces
        self.layers = {}
        self.launches = 0

    def conv_bn(self, x, conv, bn, res=None, relu=True):
        """relu?(bn(conv(x)) + res) on a MATERIALISED fp32 N,C,T,H,W tensor -> materialised tensor."""
        L = self.layers.get(id(conv))
        if L is None:
            L = self.layers[id(conv)] = _Folded(conv, bn, self.pieces)
        plan = ops.plan_for(x, conv)
        first = plan if plan.chunks is None else plan.chunks[0][2]
        if L.ok is None:
            L.ok = bool(C.slv_conv_fwd_eval_ok(first.gp)) and first.wf_elems > 0
        if not L.ok:
            # the 3 / 1-channel stems: native fp32 kernel on the ORIGINAL weights, then the affine (+ residual) + ReLU pass
            y, _, _ = ops.conv_fwd(plan, x, conv.weight, want_stats=False)
            return ops.bn_act(y, L.ss, res=res, relu=relu)
        if L.img is None:
            L.img, _ = ops.conv_w_transform(first, L.w, need_wt=False)          # once per pass (weight layouts do not depend on the batch)
        y = torch.empty(plan.out_shape, dtype=torch.float32, device=x.device)
        if plan.chunks is None:
            C.slv_conv_fwd_eval(plan.gp, ptr(x), ptr(L.img), ptr(plan.tab_fwd), ptr(L.bias), ptr(res), int(relu), self.pieces,
                                ptr(y), self._cfg(L, plan, x, y), stream())
            self.launches += 1
            return y
        for b0, b1, sub in plan.chunks:                                         # batch slices at the 32-bit buffer range
            C.slv_conv_fwd_eval(sub.gp, ptr(x[b0:b1]), ptr(L.img), ptr(sub.tab_fwd), ptr(L.bias),
                                ptr(None if res is None else res[b0:b1]), int(relu), self.pieces, ptr(y[b0:b1]),
                                self._cfg(L, sub, x[b0:b1], y[b0:b1]), stream())
            self.launches += 1
        return y

    def _cfg(self, L, plan, x, y):
        """Launch configuration of the eval kernel for this layer shape.  Benchmark mode (ops.set_benchmark, main.py:187
        cudnn.benchmark = True): every unsplit tile candidate is timed once on the layer's own tensors -- the training forward's
        tuned tile was chosen for a kernel with a load prologue, a statistics epilogue and twice the matrix-core work of the
        two-piece form; otherwise the training forward's configuration."""
        cfg = L.cfg.get(id(plan))
        if cfg is not None:
            return cfg
        cfg = plan.cfg_fwd if (plan.cfg_fwd >> 16) <= 1 else 0                  # (K is never split here)
        if ops.benchmark and x.is_cuda:
            best = None
            for cand in [cfg] + [c for c in plan.candidates(0) if (c >> 16) <= 1 and ((c >> 12) & 15) == 0 and c != cfg]:
                try:
                    t = ops._time_call(lambda: C.slv_conv_fwd_eval(plan.gp, ptr(x), ptr(L.img), ptr(plan.tab_fwd), ptr(L.bias), 0, 1,
                                                                   self.pieces, ptr(y), cand, stream()))
                except Exception:
                    continue
                if best is None or t < best[0] * 0.97:
                    best = (t, cand)
            if best is not None:
                cfg = best[1]
        L.cfg[id(plan)] = cfg
        return cfg


def _trunks(model):
    m = model.module if hasattr(model, "module") else model
    return [m.video_network.base, m.audio_network.base]


@contextlib.contextmanager
def folded_eval(model, pieces=3):
    """Inside the block every eval-mode (no-grad) forward of ``model``'s two fp32 trunks runs the folded schedule
    (engine.video_stage_forward / audio_forward consult ``trunk._folded_eval``).  The folded weights and their images are
    those of the weights AS THEY ARE when first used inside the block."""
    trunks = _trunks(model)
    fe = FoldedEval(pieces)
    for t in trunks:
        t._folded_eval = fe
    try:
        yield fe
    finally:
        for t in trunks:
            t._folded_eval = None
