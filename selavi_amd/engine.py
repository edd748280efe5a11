"""Forward/backward engine of the two trunks: an explicit schedule of HIP kernels (no autograd
inside, no per-layer torch modules executed).

Every conv of R(2+1)D-18 / ResNet-9 is followed by a BatchNorm (torchvision nets built by
/root/reference/model.py:95,114).  The engine keeps only the RAW conv outputs in HBM; BN(+ReLU) is
applied when the consumer loads the tensor (PRO_ACT), batch statistics come out of the producing
conv's epilogue, and in the backward pass BN-backward is folded into per-channel coefficients
(bwd5) with which the conv-output gradient is materialised ONCE per layer (slv_bn_bwd_apply) and then
read by both dgrad and wgrad.  In the forward pass only block outputs (two consumers) are
materialised.  See csrc/igemm.hpp.
"""
import torch

import os

from . import ops as _ops32

# BatchNorm-backward partial sums from the dgrad epilogue (True) or from a separate reduce pass (False)
FUSE_BN_BWD_REDUCE = os.environ.get("SELAVI_FUSE_BNR", "1") == "1"
# Weight gradients on a second HIP stream: they are off the critical path of the backward sweep (only the
# optimizer needs them), so their tails and the under-filled late-layer launches overlap the next dgrad.
WGRAD_SIDE_STREAM = os.environ.get("SELAVI_WGRAD_STREAM", "1") == "1"
_WGRAD_STREAMS = {}


def _wgrad_stream(device, cur):
    key = (device, cur.cuda_stream)
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def join_side_streams(ctx):
    """Called at the end of a backward schedule: the caller's stream waits for the weight gradients."""
    if ctx.side is not None:
        torch.cuda.current_stream(ctx.side.device).wait_stream(ctx.side)
        ctx.side = None


class Raw:
    """A raw conv output with its (pending) BatchNorm."""
    __slots__ = ("y", "ss", "mi", "plan", "conv", "bn", "src", "wt", "pending", "patch")

    def __init__(self, y, ss, mi, plan, conv, bn, src, wt=None):
        self.y, self.ss, self.mi, self.plan, self.conv, self.bn, self.src = y, ss, mi, plan, conv, bn, src
        self.wt = wt                # backward-data weights made together with the forward ones
        self.pending = None         # (ssum, ssq, count): statistics not finalised yet (finalize_deferred)
        self.patch = None           # 16-bit stem conv: the W-patch image of the input, kept for the weight gradient


class Ctx:
    """Execution context of one trunk pass."""

    def __init__(self, training, sync=None, ops=None):
        self.training = training
        self.ops = ops if ops is not None else _ops32      # kernel backend: selavi_amd.ops (fp32 N,C,T,H,W) or
                                                            # selavi_amd.ops16 (bf16 channels-last, fp32 master weights)
        self.sync = sync            # (process_group, world_size) for SyncBN or None
        self.grads = {}             # id(param) -> grad tensor
        self.grad_out = None        # id(param) -> preallocated gradient view (parallel.GradSink), or None
        self.wimg = None            # ops.WeightImages of this trunk pass (all weight re-layouts in one launch), if any
        self.side = None            # HIP stream carrying this pass' weight-gradient launches, if any
        self.wgrad_side = WGRAD_SIDE_STREAM      # (off for a trunk that itself runs on a side stream: nn.TrunkFunction)
        self.folded = None          # infer32.FoldedEval: eval-mode forward with BatchNorm folded into the weights, if any

    def grad_like(self, p):
        """Where the gradient of parameter p is written: the data-parallel bucket view if there is one."""
        v = self.grad_out.get(id(p)) if self.grad_out is not None else None
        return torch.empty_like(p) if v is None else v


def _as5d(t):
    return t if t.dim() == 5 else t.unsqueeze(2)


def conv_bn(ctx, x, conv, bn, need_dx=True, defer=False):
    """x: materialised tensor or Raw (then relu(bn(x)) is applied on load).  -> Raw
    need_dx=False: first conv of a trunk (no gradient w.r.t. the network input is ever taken).
    defer (SyncBN only): leave the statistics un-finalised -- finalize_deferred packs several layers into one exchange."""
    if isinstance(x, Raw):
        xin, in_ss = x.y, x.ss
    else:
        xin, in_ss = x, None
    plan = ctx.ops.plan_for(xin, conv)
    # one pass over the weights makes the forward (tap-major) and backward-data layouts of this step
    # (issuing these small kernels on the side stream was measured: no gain)
    need_wt = ctx.training and need_dx
    got = ctx.wimg.get(plan, conv.weight, need_wt) if (ctx.wimg is not None and ctx.wimg.ready) else None
    if got is not None:
        wf, wt = got                                  # made by the trunk's one launch (ops.WeightImages.run)
    else:
        wf, wt = ctx.ops.conv_w_transform(plan, conv.weight, need_wt=need_wt)
        if ctx.wimg is not None and not ctx.wimg.ready:
            ctx.wimg.note(plan, conv.weight, need_wt)
    patch = None
    if getattr(plan, "stem", False):      # (1.64 GB at 128 clips x 32 frames; making it twice cost 0.95 ms of the step)
        patch = ctx.ops.stem_patch(plan, xin)
        y, ssum, ssq = ctx.ops.conv_fwd(plan, xin, conv.weight, want_stats=ctx.training, wf=wf, patch=patch)
        if not ctx.training:
            patch = None
    else:
        y, ssum, ssq = ctx.ops.conv_fwd(plan, xin, conv.weight, in_ss=in_ss, in_relu=in_ss is not None,
                                    want_stats=ctx.training, wf=wf)
    if ctx.training and defer and ctx.sync is not None:
        r = Raw(y, None, None, plan, conv, bn, x, wt)
        r.pending = (ssum, ssq, plan.count)
        r.patch = patch
        bn.note_batch()
        return r
    if ctx.training:
        mi, ss = ctx.ops.bn_train_finalize(ssum, ssq, plan.count, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       bn.momentum, bn.eps, sync=ctx.sync)
        bn.note_batch()
    else:
        mi, ss = ctx.ops.bn_eval_params(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    r = Raw(y, ss, mi, plan, conv, bn, x, wt)
    r.patch = patch
    return r


def finalize_deferred(ctx, raws):
    """One SyncBN exchange for the deferred statistics of several layers (the main stream's latency-bound exchanges are
    what data parallelism adds to the step: main.py:117-118)."""
    raws = [r for r in raws if r is not None and r.pending is not None]
    if not raws:
        return
    items = [(*r.pending, r.bn.weight, r.bn.bias, r.bn.running_mean, r.bn.running_var, r.bn.momentum, r.bn.eps) for r in raws]
    for r, (mi, ss) in zip(raws, ctx.ops.bn_train_finalize_many(items, ctx.sync)):
        r.mi, r.ss, r.pending = mi, ss, None


def tail(ctx, r, res=None, res_raw=None, relu=True):
    """Materialise relu(bn(r) + residual)."""
    if res_raw is not None:
        return ctx.ops.bn_act(r.y, r.ss, res=res_raw.y, res_ss=res_raw.ss, relu=relu)
    return ctx.ops.bn_act(r.y, r.ss, res=res, relu=relu)


def backprop_raw(ctx, r, g, b5, a_relu, need_dx=True, addend=None, out=None, keep_g=False, fuse_bn=False, applied=False):
    """Backward through conv `r.conv` given the gradient `g` w.r.t. the ACTIVATED output of r's BN
    (or the masked tail gradient when a_relu is False) and its folded BN-backward coefficients.
    Writes the weight gradient; returns the gradient w.r.t. the conv input (activated, if the
    source is itself Raw) or None.  fuse_bn (source is Raw, consumed through its own ReLU): the dgrad
    epilogue also forms the source BN's backward partial sums -> returns (dx, part) for bn_bwd_own.
    applied: g already IS the gradient w.r.t. r's raw output (the backward-data conv that produced it applied r's BatchNorm
    backward in its epilogue).  When this conv can do the same for ITS source it returns (dx, ("applied", b5 of the source))."""
    src = r.src
    if isinstance(src, Raw):
        xin, in_ss = src.y, src.ss
    else:
        xin, in_ss = src, None
    # materialise dXout once (in place over g, which is dead afterwards unless it doubles as the
    # residual addend) and feed plain tensors to both GEMMs (1.8x faster than folding the BN backward
    # into the wgrad/dgrad operand loaders, which is what round 1 started with)
    # the stem's first conv on the 16-bit path: no backward data, and its weight gradient applies the BatchNorm backward on load
    wg_apply = (not need_dx and not applied and not fuse_bn and getattr(ctx.ops, "stem_wgrad_apply_ok", None) is not None
                and ctx.ops.stem_wgrad_apply_ok(r.plan))
    if wg_apply:
        dxo = g
    else:
        dxo = g if applied else ctx.ops.bn_bwd_apply(g, r.y, b5, a_relu, out=torch.empty_like(g) if (keep_g or addend is g) else None)

    def dgrad():
        wt = r.wt if r.wt is not None else ctx.ops.conv_wt_transform(r.plan, r.conv.weight)
        if fuse_bn and not (FUSE_BN_BWD_REDUCE and ctx.ops.FUSE_BNR):
            return ctx.ops.conv_dgrad(r.plan, dxo, wt, addend=addend, out=out), None
        bnr = (src.y, src.ss, src.mi) if fuse_bn else None
        return ctx.ops.conv_dgrad(r.plan, dxo, wt, addend=addend, out=out, bnr=bnr)

    w = r.conv.weight
    dw_out = ctx.grad_out.get(id(w)) if ctx.grad_out is not None else None     # persistent bucket view (parallel.py)
    if dw_out is not None:
        dw_out = dw_out.view(w.shape[0], -1)

    def wgrad():
        if wg_apply:
            return ctx.ops.conv_wgrad(r.plan, dxo, xin, out=dw_out, bn_apply=(r.y, b5, a_relu))
        if r.patch is not None:
            return ctx.ops.conv_wgrad(r.plan, dxo, xin, in_ss=in_ss, in_relu=in_ss is not None, out=dw_out, patch=r.patch)
        return ctx.ops.conv_wgrad(r.plan, dxo, xin, in_ss=in_ss, in_relu=in_ss is not None, out=dw_out)

    if fuse_bn and in_ss is not None and getattr(ctx.ops, "wgrad_bnr_ok", None) is not None and ctx.ops.wgrad_bnr_ok(r.plan):
        # stride-1 temporal conv on the 16-bit path: its weight gradient also yields the source BatchNorm's backward sums
        # (csrc/wgrad_cl16_t2.hip) -- on this stream, the sums are on the critical path; no reduce pass over g and x
        dw, part = ctx.ops.conv_wgrad(r.plan, dxo, xin, in_ss=in_ss, in_relu=True, out=dw_out, bnr=(src.mi, w))
        ctx.grads[id(w)] = dw.view_as(w)
        wt = r.wt if r.wt is not None else ctx.ops.conv_wt_transform(r.plan, w)
        if need_dx and addend is None and getattr(ctx.ops, "dgrad_apply_ok", None) is not None and ctx.ops.dgrad_apply_ok(r.plan):
            # the source BatchNorm's backward coefficients exist before its gradient does: the backward-data conv applies
            # them in its epilogue and stores the gradient w.r.t. the source's RAW output (no bn_bwd_apply pass)
            b5_src = bn_bwd_own(ctx, src, None, part)
            dx = ctx.ops.conv_dgrad(r.plan, dxo, wt, out=out, bn_apply=(src.y, b5_src))
            return dx, ("applied", b5_src)
        dx = ctx.ops.conv_dgrad(r.plan, dxo, wt, addend=addend, out=out) if need_dx else None
        return dx, part
    if ctx.wgrad_side:
        # the backward-data conv is on the critical path: it is enqueued first; the weight gradient starts on
        # the side stream as soon as dXout exists (event recorded before the dgrad launch)
        cur = torch.cuda.current_stream(dxo.device)
        side = _wgrad_stream(dxo.device, cur)
        ready = cur.record_event()
        res = dgrad() if need_dx else None
        side.wait_event(ready)
        with torch.cuda.stream(side):
            dw = wgrad()
        for t in (dxo, xin, in_ss, r.patch) + ((r.y, b5) if wg_apply else ()):        # allocated on `cur`, read on `side`
            if t is not None:
                t.record_stream(side)
        if dw_out is None:
            dw.record_stream(cur)                   # allocated on `side`, consumed by the optimizer on `cur`
        ctx.side = side
    else:
        dw = wgrad()
        res = dgrad() if need_dx else None
    ctx.grads[id(r.conv.weight)] = dw.view_as(r.conv.weight)
    return res


def bn_bwd_own(ctx, r, g, part=None):
    """BN backward coefficients for Raw r consumed through relu(bn(.)) with upstream gradient g
    (part: the partial sums, when the dgrad that produced g already formed them)."""
    dg, db = ctx.grad_like(r.bn.weight), ctx.grad_like(r.bn.bias)
    b5, _, _ = ctx.ops.bn_bwd(g, r.y, r.mi, r.bn.weight, ss_mask=r.ss, sync=ctx.sync, dgamma=dg, dbeta=db, part=part)
    ctx.grads[id(r.bn.weight)] = dg
    ctx.grads[id(r.bn.bias)] = db
    return b5


class BlockRec:
    __slots__ = ("u_in", "chain", "ds", "v")


def block_fwd(ctx, u, chain_mods, ds_mods):
    """Residual block: chain of conv+BN (ReLU between them), optional downsample conv+BN,
    output relu(bn(last) + shortcut).  chain_mods: [(conv, bn), ...]"""
    rec = BlockRec()
    rec.u_in = u
    x = u
    rec.chain = []
    for k, (conv, bn) in enumerate(chain_mods):
        # the last BatchNorm of the chain and the downsample BatchNorm are both consumed by the block tail only: their
        # statistics travel in one exchange
        x = conv_bn(ctx, x, conv, bn, defer=ds_mods is not None and k == len(chain_mods) - 1)
        rec.chain.append(x)
    rec.ds = conv_bn(ctx, u, ds_mods[0], ds_mods[1], defer=True) if ds_mods is not None else None
    finalize_deferred(ctx, [rec.chain[-1], rec.ds])
    if rec.ds is None and isinstance(u, Raw):              # (the un-materialised stem output as the shortcut)
        last = rec.chain[-1]
        rec.v = ctx.ops.bn_act(last.y, last.ss, res=u.y, res_ss=u.ss, relu=True, res_relu=True)
    else:
        rec.v = tail(ctx, rec.chain[-1], res=None if rec.ds is not None else u, res_raw=rec.ds)
    return rec


def block_fwd_folded(ctx, u, chain_mods, ds_mods):
    """Eval mode, BatchNorm folded into the weights (infer32.FoldedEval): every conv + BN (+ shortcut) + ReLU of the block is
    one launch on materialised tensors -- relu(bn(last) + shortcut) comes out of the last conv's epilogue."""
    f = ctx.folded
    x = u
    for conv, bn in chain_mods[:-1]:
        x = f.conv_bn(x, conv, bn, relu=True)
    shortcut = u if ds_mods is None else f.conv_bn(u, ds_mods[0], ds_mods[1], relu=False)
    return f.conv_bn(x, chain_mods[-1][0], chain_mods[-1][1], res=shortcut, relu=True)


def block_bwd(ctx, rec, dv, need_du=True):
    """dv: gradient w.r.t. the block output.  Returns gradient w.r.t. the block input."""
    last = rec.chain[-1]
    ds = rec.ds
    dg, db = ctx.grad_like(last.bn.weight), ctx.grad_like(last.bn.bias)
    if ds is not None:
        dg2, db2 = ctx.grad_like(ds.bn.weight), ctx.grad_like(ds.bn.bias)
        b5, b5ds, dz = ctx.ops.bn_bwd(dv, last.y, last.mi, last.bn.weight, v_mask=rec.v, x2=ds.y, mi2=ds.mi,
                                  gamma2=ds.bn.weight, sync=ctx.sync, dgamma=dg, dbeta=db, dgamma2=dg2, dbeta2=db2)
        ctx.grads[id(ds.bn.weight)] = dg2
        ctx.grads[id(ds.bn.bias)] = db2
    else:
        b5, b5ds, dz = ctx.ops.bn_bwd(dv, last.y, last.mi, last.bn.weight, v_mask=rec.v, sync=ctx.sync, dgamma=dg,
                                  dbeta=db)
    ctx.grads[id(last.bn.weight)] = dg
    ctx.grads[id(last.bn.bias)] = db
    g, a_relu = dz, False
    n = len(rec.chain)
    applied = False                  # g is already the gradient w.r.t. the conv's raw output (fused into the producer)
    for i in range(n - 1, -1, -1):
        r = rec.chain[i]
        if i > 0:
            g, part = backprop_raw(ctx, r, g, b5, a_relu, keep_g=(i == n - 1),  # dz is reused by the shortcut
                                   fuse_bn=True, applied=applied)
            if isinstance(part, tuple):
                b5, applied = part[1], True
            else:
                b5, applied = bn_bwd_own(ctx, rec.chain[i - 1], g, part), False
            a_relu = True
        else:
            if ds is not None:
                du = backprop_raw(ctx, r, g, b5, a_relu, need_dx=need_du, keep_g=(n == 1), applied=applied)
                du = backprop_raw(ctx, ds, dz, b5ds, False, need_dx=need_du, addend=du, out=du)
            else:
                du = backprop_raw(ctx, r, g, b5, a_relu, need_dx=need_du, addend=dz, applied=applied)
            return du


# ------------------------------------------------------------------------------------------ trunks
VIDEO_STAGES = ("stem", "layer1", "layer2", "layer3", "layer4")


def _video_blocks(layer):
    for blk in layer:
        chain = [(blk.conv1[0][0], blk.conv1[0][1]), (blk.conv1[0][3], blk.conv1[1]),
                 (blk.conv2[0][0], blk.conv2[0][1]), (blk.conv2[0][3], blk.conv2[1])]
        ds = (blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
        yield chain, ds


def video_stage_forward(ctx, base, stage, x, aux=None):
    """One stage of R(2+1)D-18 (torchvision VideoResNet): the stem, or a residual layer
    (layer4 ends with the global average pool -> feat [B,512]).  Returns (output, saved record).
    The trunk is cut into stages so that every stage is its own autograd node: under DDP the
    gradients of layer4 (75 % of the parameters) are all-reduced while layers 3..1 still run backward.
    ``aux``: what the previous stage returned beside its output -- the stem's scale / shift when its output is handed over
    un-materialised (LAZY_STEM_TAIL: the stem then returns (raw tensor, scale_shift) and layer1 takes that pair)."""
    if ctx.folded is not None and not ctx.training:
        if stage == "stem":
            st = base.stem
            return ctx.folded.conv_bn(ctx.folded.conv_bn(x, st[0], st[1]), st[3], st[4]), None
        u = x
        for chain, ds in _video_blocks(getattr(base, stage)):
            u = block_fwd_folded(ctx, u, chain, ds)
        return (ctx.ops.avgpool_fwd(u) if stage == "layer4" else u), None
    if stage == "stem":
        st = base.stem
        r0 = conv_bn(ctx, x, st[0], st[1], need_dx=False)
        r1 = conv_bn(ctx, r0, st[3], st[4])
        if getattr(ctx.ops, "LAZY_STEM_TAIL", False):
            # the stem's output relu(bn(.)) is NOT materialised: layer 1's first conv and its first block tail apply the
            # BatchNorm + ReLU on load (one pass over a 64-channel tensor less: 0.7 ms of the cfg5 forward).  The stage hands
            # the RAW tensor to the next autograd node and its scale / shift beside it as a second (non-differentiable)
            # output; the gradient that comes back is, as before, the one w.r.t. the ACTIVATED output.
            return (r1.y, r1.ss), (r0, r1)
        return tail(ctx, r1), (r0, r1)
    u, recs = x, []
    if aux is not None:
        assert stage == "layer1", "only the stem hands its output over un-materialised"
        u = Raw(x, aux, None, None, None, None, None)
    for chain, ds in _video_blocks(getattr(base, stage)):
        rec = block_fwd(ctx, u, chain, ds)
        recs.append(rec)
        u = rec.v
    if stage == "layer4":
        return ctx.ops.avgpool_fwd(u), (recs, u)
    return u, (recs, None)


def video_stage_backward(ctx, stage, saved, dout):
    """Gradient w.r.t. the stage input (None for the stem); parameter gradients land in ctx.grads."""
    if stage == "stem":
        r0, r1 = saved
        b5 = bn_bwd_own(ctx, r1, dout)
        g, part = backprop_raw(ctx, r1, dout, b5, True, keep_g=True, fuse_bn=True)   # dout belongs to autograd
        b5 = bn_bwd_own(ctx, r0, g, part)
        backprop_raw(ctx, r0, g, b5, True, need_dx=False)
        join_side_streams(ctx)
        return None
    recs, u_last = saved
    dv = ctx.ops.avgpool_bwd(dout.contiguous(), u_last) if stage == "layer4" else dout
    for rec in reversed(recs):
        dv = block_bwd(ctx, rec, dv)
    join_side_streams(ctx)       # (joining only once per trunk would gain 0.2 ms: measured, not worth the hazard)
    return dv


def video_forward(ctx, base, x):
    """Whole trunk in one call (used by tools/tests): returns (feat [B,512], saved records)."""
    saved, aux = [], None
    for st in VIDEO_STAGES:
        x, sv = video_stage_forward(ctx, base, st, x, aux)
        x, aux = x if isinstance(x, tuple) else (x, None)
        saved.append(sv)
    return x, saved


def video_backward(ctx, saved, dfeat):
    d = dfeat
    for st, sv in zip(reversed(VIDEO_STAGES), reversed(saved)):
        d = video_stage_backward(ctx, st, sv, d)


def audio_forward(ctx, base, spec):
    """ResNet-9/18/34 (two-conv blocks) and ResNet-50 (three-conv bottlenecks) on 1 x F x T' spectrograms (torchvision ResNet,
    SURVEY 8 a3, model.py:103-121), 2-D = 3-D with T=1."""
    x = _as5d(spec)
    folded = ctx.folded if not ctx.training else None
    r0 = conv_bn(ctx, x, base.conv1, base.bn1, need_dx=False)
    u, idx = ctx.ops.bnrelu_maxpool_fwd(r0.y, r0.ss)
    recs = []
    if folded is not None:
        for layer in (base.layer1, base.layer2, base.layer3, base.layer4):
            for blk in layer:
                chain = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)] + ([(blk.conv3, blk.bn3)] if hasattr(blk, "conv3") else [])
                ds = (blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                u = block_fwd_folded(ctx, u, chain, ds)
        return ctx.ops.avgpool_fwd(u), None
    for layer in (base.layer1, base.layer2, base.layer3, base.layer4):
        for blk in layer:
            chain = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
            if hasattr(blk, "conv3"):
                chain.append((blk.conv3, blk.bn3))
            ds = (blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
            rec = block_fwd(ctx, u, chain, ds)
            recs.append(rec)
            u = rec.v
    feat = ctx.ops.avgpool_fwd(u)
    return feat, (x, r0, idx, recs, u)


def audio_backward(ctx, saved, dfeat):
    x, r0, idx, recs, u_last = saved
    dv = ctx.ops.avgpool_bwd(dfeat.contiguous(), u_last)
    for rec in reversed(recs):
        dv = block_bwd(ctx, rec, dv)
    dy0 = ctx.ops.maxpool_bwd(dv, idx, tuple(r0.y.shape))
    b5 = bn_bwd_own(ctx, r0, dy0)
    backprop_raw(ctx, r0, dy0, b5, True, need_dx=False)
    join_side_streams(ctx)
