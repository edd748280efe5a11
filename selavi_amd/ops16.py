"""Host side of the 16-bit MFMA path (csrc/conv_cl16*.hip, train_cl16.hip, wgrad_cl16*.hip): eval forward
(``Conv16`` / infer16.py) AND training (``plan_for`` / ``conv_fwd`` / ``conv_dgrad`` / ``conv_wgrad`` / BatchNorm kernels, the
backend ``engine.Ctx.ops`` of ``AVModel.set_precision("bf16")``).  Opt-in: every bit-exactness claim stays with fp32.

Activations are bf16 channels-last ``[N, T, H, W, Cp]`` torch tensors (Cp = channels rounded up to 32, padding
channels zero); weights are re-laid-out once per layer (``Conv16.from_weight``).  The epilogue applies a per-channel
affine (eval-mode BatchNorm), an optional residual and ReLU, i.e. one launch per conv+BN(+add)+ReLU of a block.
"""
import numpy as np
import torch

from ._lib import C, ptr, stream


def pad32(c):
    return (c + 31) // 32 * 32


def to_channels_last16(x):
    """fp32 N,C,T,H,W (or N,C,H,W) -> bf16 N,T,H,W,Cp."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    x = x.contiguous()
    N, Cc, T, H, W = x.shape
    y = torch.empty((N, T, H, W, pad32(Cc)), dtype=torch.bfloat16, device=x.device)
    C.slv_to_cl16(ptr(x), ptr(y), N, Cc, pad32(Cc), T * H * W, stream())
    return y


def _pick_mt(cout):
    best = None
    for mt in (9, 8, 4):
        rows = -(-cout // (16 * mt)) * 16 * mt
        if best is None or rows < best[1]:
            best = (mt, rows)
    # the 8-wave kernel of the wide layers (csrc/conv_cl16_g8.hip) takes weight layouts whose rows come in blocks of 288, 256
    # or 128: 921 channels -> 1 024 rows instead of 960 (7 % more rows, one kernel for the whole layer)
    if cout > 256 and best[1] % 128 and best[1] % 288:
        rows = -(-cout // 256) * 256
        if rows <= 1.08 * best[1]:
            best = (8, rows)
    return best


class Conv16:
    """One conv layer (weights [Cout, Cin, kt, kh, kw] fp32, 2-D convs: kt = 1) prepared for slv_conv_cl16_fwd."""

    def __init__(self, w, stride, pad):
        if w.dim() == 4:
            w = w.unsqueeze(2)
        self.Cout, self.Cin, self.k = w.shape[0], w.shape[1], tuple(w.shape[2:])
        self.stride, self.pad = tuple(stride), tuple(pad)
        self.Cin_p, self.Cout_p = pad32(self.Cin), pad32(self.Cout)
        self.mt, self.Mrows = _pick_mt(self.Cout)
        taps = self.k[0] * self.k[1] * self.k[2]
        wl = torch.zeros((taps, self.Cin_p // 32, self.Mrows, 32), dtype=torch.float32, device=w.device)
        wp = torch.zeros((self.Cout, self.Cin_p, taps), dtype=torch.float32, device=w.device)
        wp[:, :self.Cin] = w.reshape(self.Cout, self.Cin, taps)
        # [Cout][Cin_p/32][32][taps] -> [taps][Cin_p/32][Cout][32]
        wl[:, :, :self.Cout] = wp.reshape(self.Cout, self.Cin_p // 32, 32, taps).permute(3, 1, 0, 2)
        self.wl = wl.to(torch.bfloat16).contiguous()

    def out_shape(self, x):
        N, T, H, W, _ = x.shape
        (kt, kh, kw), (st, sh, sw), (pt, ph, pw) = self.k, self.stride, self.pad
        return N, (T + 2 * pt - kt) // st + 1, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1, self.Cout_p

    def __call__(self, x, scale_shift=None, res=None, relu=False):
        assert x.dtype == torch.bfloat16 and x.dim() == 5 and x.shape[4] == self.Cin_p and x.is_contiguous()
        shp = self.out_shape(x)
        y = torch.empty(shp, dtype=torch.bfloat16, device=x.device)
        if res is not None:
            assert res.shape == y.shape and res.dtype == torch.bfloat16 and res.is_contiguous()
        N, T, H, W, _ = x.shape
        g = np.array([N, T, H, W, self.Cin_p, self.Cout, self.Cout_p, shp[1], shp[2], shp[3], *self.k, *self.stride,
                      *self.pad, self.Mrows], dtype=np.int32)
        C.slv_conv_cl16_fwd(g.ctypes.data, self.mt, ptr(x), ptr(self.wl), ptr(y), ptr(scale_shift), ptr(res), int(relu),
                            stream())
        return y


class StemConv16:
    """A (1, kh, kw) stem conv over C <= 4 input channels (video: 3 -> 45, (1,7,7); audio: 1 -> 64, 7x7) as a
    (1, kh, 1) conv over the 32-channel W-patch layout of slv_to_cl16_wpatch: kh K-steps instead of kh*kw."""

    def __init__(self, w, stride, pad):
        if w.dim() == 4:
            w = w.unsqueeze(2)
        Cout, Cin, kt, kh, kw = w.shape
        assert kt == 1 and stride[0] == 1 and pad[0] == 0 and kw * Cin <= 32
        self.Cin, self.kw, self.sw, self.pw = Cin, kw, stride[2], pad[2]
        w2 = torch.zeros((Cout, 32, 1, kh, 1), dtype=torch.float32, device=w.device)
        # patch channel dw*Cin + c  <-  w[:, c, 0, :, dw]
        w2[:, :kw * Cin, 0, :, 0] = w[:, :, 0].permute(0, 3, 1, 2).reshape(Cout, kw * Cin, kh)
        self.conv = Conv16(w2, (1, stride[1], 1), (0, pad[1], 0))

    def __call__(self, x, scale_shift=None, relu=False):
        """x: fp32 N,C,T,H,W (or N,C,H,W)."""
        if x.dim() == 4:
            x = x.unsqueeze(2)
        x = x.contiguous()
        N, Cc, T, H, W = x.shape
        assert Cc == self.Cin
        Wo = (W + 2 * self.pw - self.kw) // self.sw + 1
        p = torch.empty((N, T, H, Wo, 32), dtype=torch.bfloat16, device=x.device)
        C.slv_to_cl16_wpatch(ptr(x), ptr(p), N, Cc, T * H, W, self.kw, self.sw, self.pw, stream())
        return self.conv(p, scale_shift=scale_shift, relu=relu)


# ======================================================================================================================
# Training on the 16-bit path (BASELINE configs[4]; /root/reference/main.py:151-153 apex O1, :296-299).  The functions
# below have the signatures of their fp32 counterparts in selavi_amd/ops.py, so selavi_amd/engine.py runs the same
# forward/backward schedule on either backend (engine.Ctx.ops).  Activations and activation gradients are bf16
# channels-last [N, T, H, W, Cp]; weights, BatchNorm parameters, statistics and weight gradients stay fp32.
# ======================================================================================================================
import os

from . import ops as _ops

# BatchNorm-backward sums from the backward-data epilogue (conv_dgrad(bnr=...), EPI 2): implemented and tested, OFF by
# default -- in the step the epilogue's reads of x in accumulator layout (8-byte pieces, one row per lane) cost the
# layer-1 backward-data launches +300 us against the 45-110 us of the separate coalesced reduce pass it replaces
# (profiles/r02_notes.md); SELAVI_CL16_FUSE_BNR=1 switches it on.
FUSE_BNR = os.environ.get("SELAVI_CL16_FUSE_BNR", "0") == "1"
# engine.video_stage_forward: the stem's block output is applied on load by its consumers instead of being materialised
LAZY_STEM_TAIL = os.environ.get("SELAVI_CL16_LAZY_STEM", "1") == "1"
# ... with the BatchNorm-backward apply of the stem's first conv folded into its loader (no bn_bwd_apply pass over that tensor)
STEM_WGRAD_APPLY = os.environ.get("SELAVI_CL16_STEM_WGRAD_APPLY", "1") == "1"
# the stem's weight gradient on the direct kernel as well (then no W-patch tensor exists in the step)
STEM_DIRECT_WGRAD = os.environ.get("SELAVI_CL16_STEM_WGRAD", "1") == "1"
bn_train_finalize = _ops.bn_train_finalize
bn_train_finalize_many = _ops.bn_train_finalize_many
bn_eval_params = _ops.bn_eval_params


def bnrelu_maxpool_fwd(x, ss):
    """MaxPool2d(3, 2, 1)(relu(bn(x))) of the audio stem on bf16 [N,1,H,W,Cp]: (out, idx)."""
    N, T, H, W, Cp = x.shape
    assert T == 1 and x.dtype == torch.bfloat16 and x.is_contiguous()
    Cc = ss.shape[1]
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = _bf16(N, 1, Ho, Wo, Cp, device=x.device)
    idx = torch.empty(N, 1, Ho, Wo, Cp, dtype=torch.uint8, device=x.device)
    C.slv_cl16_bnrelu_maxpool_fwd(ptr(x), ptr(ss), ptr(out), ptr(idx), N, Cc, Cp, H, W, stream())
    return out, idx


def maxpool_bwd(dout, idx, in_shape):
    N, T, H, W, Cp = in_shape
    assert T == 1 and dout.dtype == torch.bfloat16 and dout.is_contiguous()
    dy = _bf16(*in_shape, device=dout.device)
    C.slv_cl16_maxpool_bwd(ptr(dout), ptr(idx), ptr(dy), N, Cp, H, W, stream())
    return dy


CL_BUF_LIMIT = int(os.environ.get("SELAVI_CL16_BUF_LIMIT", str(0xFFFFFFF0)))
_CLC_WORDS = None


def _clconv(N, Bdims, Cin_p, Cin, L, bm, bo, Odims, Cout, Cout_p, om, oo, Mrows, taps, forward=False):
    """int32 image of csrc/conv_cl16.hip:ClConv.  taps: [(dt, dh, dw, slab)]."""
    global _CLC_WORDS
    if _CLC_WORDS is None:
        _CLC_WORDS = C.slv_cl16_conv_words()
    g = np.zeros(_CLC_WORDS, dtype=np.int32)
    assert len(taps) <= 64 and all(-8 <= d <= 7 for t in taps for d in t[:3])
    g[:28] = [N, *Bdims, Cin_p, Cin, *L, *bm, *bo, *Odims, Cout, Cout_p, *om, *oo, Mrows, len(taps)]
    for i, (dt, dh, dw, slab) in enumerate(taps):
        g[28 + i] = (dt + 8) | (dh + 8) << 4 | (dw + 8) << 8 | slab << 12
    g[28 + 64] = 1 if forward else 0          # flags: bit 0 = forward launch (the library's kernel choice)
    return g


def _pick_w(n):
    """Weight-gradient tile width (32*w, w in 2..5) for n rows/columns: least padding, then the wider tile."""
    best = None
    for w_ in (5, 4, 3, 2):
        pad = -(-n // (32 * w_)) * 32 * w_
        if best is None or pad < best[1]:
            best = (w_, pad)
    return best[0]


_flags_reconciled = False


def _reconcile_flags():
    """Before the first plan is built: the opt-in SELAVI_CL16_FUSE_BNR=1 (BatchNorm-backward sums from the backward-data
    epilogue) needs launches the 8-wave kernel's default dispatch would take away (layer-4 backward data, no fused sums
    there: slv_cl16_conv returns an error) -- the 8-wave kernel is switched off for the process then, loudly."""
    global _flags_reconciled
    if _flags_reconciled:
        return
    _flags_reconciled = True
    if FUSE_BNR and C.slv_cl16_g8_mode(-1) != 0:
        import sys
        C.slv_cl16_g8_mode(0)
        sys.stderr.write("selavi_amd.ops16: SELAVI_CL16_FUSE_BNR=1 -> the 8-wave conv kernel (SELAVI_CL16_G8) is off for this "
                         "process: it has no fused BatchNorm-backward sums\n")


class Plan16:
    """Geometry of one conv layer on the bf16 path for a given input shape: forward, backward-data (one launch per
    stride-parity class) and weight-gradient launch descriptions.  ``stem``: the (1, kh, kw) conv over <= 4 input
    channels that reads the fp32 N,C,T,H,W clip through the W-patch layout (StemConv16)."""

    _cache = {}

    @classmethod
    def get(cls, N, Ti, Hi, Wi, Cin, Cout, k, stride, pad, device, stem=False):
        key = (N, Ti, Hi, Wi, Cin, Cout, tuple(k), tuple(stride), tuple(pad), str(device), stem)
        p = cls._cache.get(key)
        if p is None:
            p = cls._cache[key] = cls(N, Ti, Hi, Wi, Cin, Cout, tuple(k), tuple(stride), tuple(pad), device, stem)
        return p

    def __init__(self, N, Ti, Hi, Wi, Cin, Cout, k, stride, pad, device, stem):
        _reconcile_flags()
        self.device, self.stem = device, stem
        self.w_shape_taps = k[0] * k[1] * k[2]
        self.Cin_w, self.Cout = Cin, Cout                  # channel counts of the fp32 weight tensor
        self.patch_kw = 0
        self.stem_direct = False
        if stem:                                           # (1,kh,kw) over Cin -> (1,kh,1) over the 32 patch channels
            assert k[0] == 1 and stride[0] == 1 and pad[0] == 0 and k[2] * Cin <= 32
            self.patch_kw, self.patch_sw, self.patch_pw = k[2], stride[2], pad[2]
            self.src_shape = (N, Cin, Ti, Hi, Wi)
            # the direct kernels (csrc/conv_cl16_stem.hip): the fp32 clip staged in LDS band by band, no W-patch tensor
            self.stem_direct = bool(C.slv_cl16_stem_ok(N, Cin, Ti, Hi, Wi, Cout, k[1], k[2], stride[1], stride[2], pad[1], pad[2]))
            self.stem_nblk = C.slv_cl16_stem_nblk(N, Cin, Ti, Hi, Wi, Cout) if self.stem_direct else 0
            Wi = (Wi + 2 * pad[2] - k[2]) // stride[2] + 1
            Cin = k[2] * Cin
            k, stride, pad = (1, k[1], 1), (1, stride[1], 1), (0, pad[1], 0)
        (kt, kh, kw), (st, sh, sw), (pt, ph, pw) = k, stride, pad
        To, Ho, Wo = (Ti + 2 * pt - kt) // st + 1, (Hi + 2 * ph - kh) // sh + 1, (Wi + 2 * pw - kw) // sw + 1
        self.N, self.Cin, self.Cin_p, self.Cout_p = N, Cin, pad32(Cin), pad32(Cout)
        self.in_dims, self.out_dims, self.k, self.stride, self.pad = (Ti, Hi, Wi), (To, Ho, Wo), k, stride, pad
        self.taps = kt * kh * kw
        self.in_shape = (N, Ti, Hi, Wi, self.Cin_p)
        self.out_shape = (N, To, Ho, Wo, self.Cout_p)
        self.count = float(N * To * Ho * Wo)
        self.mt_f, self.mrows_f = _pick_mt(Cout)
        self.mt_d, self.mrows_d = _pick_mt(self.Cin)
        self.wf_elems = self.taps * (self.Cin_p // 32) * self.mrows_f * 32
        self.wt_elems = self.taps * (self.Cout_p // 32) * self.mrows_d * 32
        # batch slices when a tensor leaves the 32-bit buffer range (clips are independent)
        per_clip = 2 * max(Ti * Hi * Wi * self.Cin_p, To * Ho * Wo * self.Cout_p)
        n_slices = -(-N // max(1, (CL_BUF_LIMIT - 1) // per_clip))
        self.chunks = None
        if n_slices > 1:
            assert not stem
            base, rem = divmod(N, n_slices)
            self.chunks, b0 = [], 0
            for i in range(n_slices):
                sz = base + (1 if i < rem else 0)
                self.chunks.append((b0, b0 + sz, Plan16.get(sz, Ti, Hi, Wi, Cin, Cout, k, stride, pad, device)))
                b0 += sz
            self.nblk = sum(c[2].nblk for c in self.chunks)
            return
        # ---- forward
        taps = [(a, b, c, (a * kh + b) * kw + c) for a in range(kt) for b in range(kh) for c in range(kw)]
        self.g_fwd = _clconv(N, (Ti, Hi, Wi), self.Cin_p, self.Cin, (To, Ho, Wo), stride, (-pt, -ph, -pw), (To, Ho, Wo),
                             Cout, self.Cout_p, (1, 1, 1), (0, 0, 0), self.mrows_f, taps, forward=True)
        self.nblk = C.slv_cl16_conv_nblk(self.g_fwd.ctypes.data)
        # ---- backward data: per dimension, class c of the input coordinate and its taps (offset, j)
        def classes(X, kk, s, p):
            out = []
            for c in range(s):
                L = -(-(X - c) // s)
                if L <= 0:
                    continue
                out.append((c, L, [((c + p - j) // s, j) for j in range(kk) if (c + p - j) % s == 0]))
            return out
        self.g_dgrad = []
        for ct, Lt, tt in classes(Ti, kt, st, pt):
            for ch, Lh, th in classes(Hi, kh, sh, ph):
                for cw, Lw, tw in classes(Wi, kw, sw, pw):
                    tp = [(a, b, c, (ja * kh + jb) * kw + jc) for a, ja in tt for b, jb in th for c, jc in tw]
                    self.g_dgrad.append(_clconv(N, (To, Ho, Wo), self.Cout_p, Cout, (Lt, Lh, Lw), (1, 1, 1), (0, 0, 0),
                                                (Ti, Hi, Wi), self.Cin, self.Cin_p, (st, sh, sw), (ct, ch, cw),
                                                self.mrows_d, tp))
        self.dgrad_nblk = [C.slv_cl16_conv_nblk(g_.ctypes.data) for g_ in self.g_dgrad]     # position tiles per class
        # backward data with the source layer's BatchNorm-backward apply in its epilogue (conv_dgrad(bn_apply=...))
        self.dgrad_apply_ok = len(self.g_dgrad) == 1 and bool(C.slv_cl16_conv_dgrad_bn_apply_ok(self.g_dgrad[0].ctypes.data))
        self.bnr_slots = sum(self.dgrad_nblk)
        # ---- weight gradient: M = Cout_p, N = taps * Cin_p, K = output positions
        self.wm, self.wn = _pick_w(self.Cout_p), _pick_w(self.taps * self.Cin_p)
        mtiles, ntiles = -(-self.Cout_p // (32 * self.wm)), -(-(self.taps * self.Cin_p) // (32 * self.wn))
        P = N * To * Ho * Wo
        ksl = max(1, min(-(-768 // (mtiles * ntiles)), P // 2048))     # >= 64 K-steps per slice: the reduce reads every slice
        kper = -(-(-(-P // ksl)) // 32) * 32
        ksl = -(-P // kper)
        self.g_wgrad = np.array([N, Ti, Hi, Wi, self.Cin_p, self.Cin, To, Ho, Wo, self.Cout_p, st, sh, sw, pt, ph, pw,
                                 kt, kh, kw, self.taps * self.Cin_p, mtiles, ntiles, ksl, kper], dtype=np.int32)
        assert len(self.g_wgrad) == C.slv_cl16_wgrad_words()
        self.ws_wgrad = C.slv_cl16_wgrad_ws_bytes(self.g_wgrad.ctypes.data, self.wm, self.wn)
        # stride-1 (3,1,1) layers: the weight gradient that also yields the BatchNorm-backward sums of the layer it reads
        self.ws_wgrad_bnr = 0 if stem else C.slv_cl16_wgrad_bnr_ws_bytes(self.g_wgrad.ctypes.data)


def plan_for(xin, conv):
    """Plan of the nn.Conv holder ``conv`` on ``xin``: bf16 [N,T,H,W,Cp], or the fp32 N,C,T,H,W clip for the stem."""
    if xin.dtype == torch.float32:
        N, Cc, T, H, W = xin.shape
        return Plan16.get(N, T, H, W, Cc, conv.out_channels, conv.kernel3, conv.stride3, conv.padding3, xin.device,
                          stem=True)
    N, T, H, W, Cp = xin.shape
    assert xin.dtype == torch.bfloat16 and Cp == pad32(conv.in_channels)
    return Plan16.get(N, T, H, W, conv.in_channels, conv.out_channels, conv.kernel3, conv.stride3, conv.padding3,
                      xin.device)


def _bf16(*shape, device):
    return torch.empty(*shape, dtype=torch.bfloat16, device=device)


def _patch(plan, x):
    N, Cc, T, H, W = plan.src_shape
    p = _bf16(*plan.in_shape, device=x.device)
    C.slv_to_cl16_wpatch(ptr(x.contiguous()), ptr(p), N, Cc, T * H, W, plan.patch_kw, plan.patch_sw, plan.patch_pw,
                         stream())
    return p


def conv_w_transform(plan, w, need_wf=True, need_wt=True):
    """fp32 master weights -> the bf16 layouts of this step (one launch): (wf, wt)."""
    need_wt = need_wt and not plan.stem
    if plan.stem and plan.stem_direct and STEM_DIRECT_WGRAD:
        return None, None                             # (the direct kernels read the fp32 master weights themselves)
    wf = _bf16(plan.wf_elems, device=w.device) if need_wf else None
    wt = _bf16(plan.wt_elems, device=w.device) if need_wt else None
    if wf is not None or wt is not None:
        C.slv_cl16_w_transform(ptr(w), ptr(wf), ptr(wt), plan.Cout, plan.Cin_w, plan.w_shape_taps if not plan.stem
                               else plan.taps, plan.Cin_p, plan.Cout_p, plan.mrows_f, plan.mrows_d, plan.patch_kw, stream())
    return wf, wt


def conv_wt_transform(plan, w):
    return conv_w_transform(plan, w, need_wf=False)[1]


class WeightImages:
    """ops.WeightImages for this backend: the bf16 weight layouts of a whole trunk in ONE launch
    (slv_cl16_w_transform_jobs) instead of one per conv layer."""

    JOB_ELEMS = 131072

    def __init__(self):
        self.rec, self.seen = [], set()
        self.ent = None
        self.table, self.njobs, self.blocks = None, 0, 0

    def note(self, plan, w, need_wt):
        if id(w) not in self.seen:
            self.seen.add(id(w))
            self.rec.append((plan, w, need_wt))

    def build(self, device):
        jobs, ent = [], {}
        for plan, w, need_wt in self.rec:
            need_wt = need_wt and not plan.stem
            wf = _bf16(plan.wf_elems, device=device)
            wt = _bf16(plan.wt_elems, device=device) if need_wt else None
            job = np.zeros(18, dtype=np.int32)
            job[0:6] = np.array([ptr(w), ptr(wf), ptr(wt)], dtype=np.uint64).view(np.int32)
            job[6:14] = [plan.Cout, plan.Cin_w, plan.w_shape_taps if not plan.stem else plan.taps, plan.Cin_p, plan.Cout_p,
                         plan.mrows_f, plan.mrows_d, plan.patch_kw]
            total = wf.numel() + (wt.numel() if wt is not None else 0)
            job[14:16] = np.array([wf.numel(), total - wf.numel()], dtype=np.uint32).view(np.int32)
            for first in range(0, total, self.JOB_ELEMS):      # equal jobs: layer 4's layouts are 30 x layer 1's
                jb = job.copy()
                jb[16:18] = np.array([first, min(self.JOB_ELEMS, total - first)], dtype=np.uint32).view(np.int32)
                jobs.append(jb)
            ent[id(w)] = (plan, wf, wt, w)
        self.rec = None
        self.ent = ent
        if jobs:
            self.table = torch.from_numpy(np.concatenate(jobs)).to(device)
            self.njobs = len(jobs)
            self.blocks = 32
        return self

    @property
    def ready(self):
        return self.ent is not None

    def run(self):
        if self.table is not None:
            C.slv_cl16_w_transform_jobs(ptr(self.table), self.njobs, self.blocks, stream())

    def get(self, plan, w, need_wt):
        e = self.ent.get(id(w)) if self.ent is not None else None
        need_wt = need_wt and not plan.stem
        if e is None or e[0] is not plan or e[3] is not w or (need_wt and e[2] is None):
            return None
        return e[1], (e[2] if need_wt else None)


def stem_patch(plan, x):
    """The W-patch image of the fp32 clip / spectrogram a stem conv reads (made once per step: the forward and the weight
    gradient both take it through ``patch=``).  None on the direct kernels (Plan16.stem_direct), which read the clip itself."""
    return _patch(plan, x) if plan.stem and not (plan.stem_direct and STEM_DIRECT_WGRAD) else None


def conv_fwd(plan, x, w, in_ss=None, in_relu=False, want_stats=True, wf=None, out=None, patch=None):
    """y = conv(relu(x*s+h)) on the MFMA kernel; returns (y, stat_sum, stat_sq) with [Cout][nblk] partials."""
    assert (in_ss is not None) == bool(in_relu), "the load prologue is BatchNorm + ReLU"
    if plan.stem and plan.stem_direct:
        assert in_ss is None and plan.chunks is None
        y = out if out is not None else _bf16(*plan.out_shape, device=x.device)
        ssum = ssq = None
        if want_stats:
            ssum = torch.empty(plan.Cout, plan.stem_nblk, dtype=torch.float32, device=x.device)
            ssq = torch.empty_like(ssum)
        N, Cc, T, H, W = plan.src_shape
        C.slv_cl16_stem_fwd(ptr(x.contiguous()), ptr(w.contiguous()), ptr(y), ptr(ssum), ptr(ssq), N, Cc, T, H, W, plan.Cout,
                            stream())
        return y, ssum, ssq
    if wf is None:
        wf, _ = conv_w_transform(plan, w, need_wt=False)
    if plan.stem:
        x = patch if patch is not None else _patch(plan, x)
    if plan.chunks is not None:
        y = _bf16(*plan.out_shape, device=x.device)
        parts = [conv_fwd(sub, x[b0:b1], w, in_ss, in_relu, want_stats, wf, out=y[b0:b1]) for b0, b1, sub in plan.chunks]
        if not want_stats:
            return y, None, None
        return y, torch.cat([p_[1] for p_ in parts], 1), torch.cat([p_[2] for p_ in parts], 1)
    y = out if out is not None else _bf16(*plan.out_shape, device=x.device)
    ssum = ssq = None
    if want_stats:
        if os.environ.get("SELAVI_DIAG_ZERO_STATS") == "1":          # diagnostic: are there partial slots no block writes?
            ssum = torch.zeros(plan.Cout, plan.nblk, dtype=torch.float32, device=x.device)
            ssq = torch.zeros_like(ssum)
        else:
            ssum = torch.empty(plan.Cout, plan.nblk, dtype=torch.float32, device=x.device)
            ssq = torch.empty_like(ssum)
    ev = _probe["plans"].get(id(plan)) if (_probe is not None and (in_ss is not None) == _probe["prologue"]) else None
    if ev is not None:       # bench.py: HIP events around THIS launch, on the stream it runs on, inside the training step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.slv_cl16_conv(plan.g_fwd.ctypes.data, plan.mt_f, ptr(x), ptr(wf), ptr(y), ptr(in_ss), 0, 0, 0, ptr(ssum), ptr(ssq),
                    0, 0, 0, 0, 0, 0, stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1))
    return y, ssum, ssq


_probe = None


class probe_conv_fwd:
    """``with ops16.probe_conv_fwd({"name": plan, ...}, prologue=True) as p: step()`` -- every training-forward launch of the
    named plans (with / without the BatchNorm + ReLU load prologue) inside the block is bracketed by HIP events on the stream it
    runs on; ``p.ms()`` -> {name: [durations]} after a synchronize.  Measurement only (bench.py's live in-step figures of
    the cfg5 leg, the counterpart of ops.probe_conv_fwd): two event records per probed launch."""

    def __init__(self, plans, prologue=True):
        self.names = {id(pl): k for k, pl in plans.items()}
        self.state = dict(plans={id(pl): [] for pl in plans.values()}, prologue=prologue)

    def __enter__(self):
        global _probe
        _probe = self.state
        return self

    def __exit__(self, *exc):
        global _probe
        _probe = None

    def ms(self):
        torch.cuda.synchronize()
        return {self.names[k]: [a.elapsed_time(b) for a, b in v] for k, v in self.state["plans"].items()}


# the BatchNorm-backward apply in the epilogue of the backward-data conv that produces the gradient (conv_dgrad(bn_apply=...),
# csrc/conv_cl16_tr.hip EPI 3): bit-exact against the separate pass; the x pieces are requested in front of the step's MFMAs
# (requested inside the store loop: one memory round trip each, 3.83 ms per launch = no gain).  cfg5 step 122.3 -> 114.6 ms.
# SELAVI_CL16_DGRAD_APPLY=0 switches it off.
DGRAD_APPLY = os.environ.get("SELAVI_CL16_DGRAD_APPLY", "1") == "1"


def dgrad_apply_ok(plan):
    """Does conv_dgrad(..., bn_apply=...) take this layer (and is it switched on)?"""
    return DGRAD_APPLY and plan.chunks is None and getattr(plan, "dgrad_apply_ok", False)


def conv_dgrad(plan, dy, wt, x_out=None, bwd5=None, relu=False, addend=None, out=None, bnr=None, bn_apply=None):
    """dx = conv_transpose(dy) (+ addend): the forward kernel on the transposed weights, one launch per parity class.
    bnr = (x, scale_shift, mean_invstd) of the layer that produced this conv's input: the epilogue also emits that
    BatchNorm's backward partial sums and (dx, part) is returned -- pass ``part`` to bn_bwd.
    bn_apply = (x, bwd5) of that layer when its BatchNorm-backward coefficients are known already (conv_wgrad(bnr=...)): the
    epilogue stores bn_bwd_apply(dx, x, bwd5, relu=True) -- the gradient w.r.t. that layer's RAW output -- bit for bit what
    the separate pass makes of the stored dx."""
    assert bwd5 is None and not plan.stem
    dx = out if out is not None else _bf16(*plan.in_shape, device=dy.device)
    if bn_apply is not None:
        assert plan.chunks is None and plan.dgrad_apply_ok and addend is None and bnr is None
        sx, b5 = bn_apply
        assert sx.shape == dx.shape and b5.shape == (5, plan.Cin)
        C.slv_cl16_conv_dgrad_bn_apply(plan.g_dgrad[0].ctypes.data, ptr(dy), ptr(wt), ptr(dx), ptr(sx), ptr(b5), stream())
        return dx
    if plan.chunks is not None:
        parts = []
        for b0, b1, sub in plan.chunks:
            r = conv_dgrad(sub, dy[b0:b1], wt, addend=None if addend is None else addend[b0:b1], out=dx[b0:b1],
                           bnr=None if bnr is None else (bnr[0][b0:b1], bnr[1], bnr[2]))
            parts.append(r[1] if bnr is not None else None)
        return dx if bnr is None else (dx, torch.cat(parts, 1))
    in_place = addend is not None and addend.data_ptr() == dx.data_ptr()
    part = rx = rss = rmi = None
    if bnr is not None:
        rx, rss, rmi = bnr
        assert rx.shape == dx.shape and rx.dtype == torch.bfloat16
        part = torch.empty(plan.Cin, plan.bnr_slots, 2, dtype=torch.float32, device=dy.device)
    slot = 0
    for g, nb in zip(plan.g_dgrad, plan.dgrad_nblk):
        if g[27] == 0 and in_place and bnr is None:   # a parity class no tap reaches (1x1x1 stride-2 downsample: 7 of 8)
            continue                                  # with dx = addend in place: nothing to do
        C.slv_cl16_conv(g.ctypes.data, plan.mt_d, ptr(dy), ptr(wt), ptr(dx), 0, 0, ptr(addend), 0, 0, 0, ptr(rx), ptr(rss),
                        ptr(rmi), ptr(part), slot, plan.bnr_slots, stream())
        slot += nb
    return dx if bnr is None else (dx, part)


# BatchNorm-backward sums from the temporal convs' weight gradient (conv_wgrad(bnr=...), csrc/wgrad_cl16_t2.hip): ON for the
# layers the accumulator-resident two-product kernel takes (csrc/wgrad_cl16_tacc.hip: the layer-1 shape 144 -> 64 with
# >= 1 024 pixel columns; cfg5 step 124.8 -> 123.4 ms).  On the column-order kernel the two products cost more than the
# reduce pass they replace (1.24 -> 4.15 ms per layer-1 launch): SELAVI_CL16_WGT2=all admits those layers too (tests).
# SELAVI_CL16_WGRAD_BNR=0 switches the path off.
WGRAD_BNR = os.environ.get("SELAVI_CL16_WGRAD_BNR", "1") == "1"


def wgrad_bnr_available(plan):
    """Does conv_wgrad(..., bnr=...) take this layer (stride-1 (3,1,1), one batch slice)?"""
    return plan.chunks is None and getattr(plan, "ws_wgrad_bnr", 0) > 0


def wgrad_bnr_ok(plan):
    """... and is it the engine's choice (SELAVI_CL16_WGRAD_BNR)?"""
    return WGRAD_BNR and wgrad_bnr_available(plan)


def stem_wgrad_apply_ok(plan):
    """Does conv_wgrad(..., bn_apply=...) take this layer (the direct stem weight gradient)?"""
    return bool(plan.stem and plan.stem_direct and STEM_DIRECT_WGRAD and STEM_WGRAD_APPLY)


def conv_wgrad(plan, dy, x_in, x_out=None, bwd5=None, a_relu=False, in_ss=None, in_relu=False, out=None, patch=None,
               bnr=None, bn_apply=None):
    """dw (fp32, [Cout][Cin*taps] = the reference layout flattened) from bf16 dy and act(x_in).
    bnr = (mean_invstd of the BatchNorm behind in_ss, this conv's fp32 weights): the kernel forms the weight gradient from
    the gradients against the masked raw activation and against the mask, which also gives that BatchNorm's backward
    sums -> returns (dw, part [Cin][1][2]) for bn_bwd(part=...), no reduce pass over the gradient (csrc/wgrad_cl16_t2.hip)."""
    assert bwd5 is None and (in_ss is not None) == bool(in_relu)
    n_w = plan.Cin_w * plan.w_shape_taps
    dw = out if out is not None else torch.empty(plan.Cout, n_w, dtype=torch.float32, device=dy.device)
    if bnr is not None:
        assert wgrad_bnr_available(plan) and in_ss is not None
        mi, w = bnr
        part = torch.empty(plan.Cin, 1, 2, dtype=torch.float32, device=dy.device)
        ws = _ops.workspace(plan.ws_wgrad_bnr, dy.device)
        C.slv_cl16_wgrad_bnr(plan.g_wgrad.ctypes.data, ptr(dy), ptr(x_in), ptr(in_ss), ptr(mi), ptr(w.contiguous()), ptr(dw),
                             ptr(part), plan.Cout, ptr(ws), plan.ws_wgrad_bnr, stream())
        return dw, part
    if plan.stem and plan.stem_direct and STEM_DIRECT_WGRAD and patch is None:
        # bn_apply = (y, bwd5, relu): dy is the gradient w.r.t. the activated output of this conv's BatchNorm; the kernel
        # applies the BatchNorm backward on load (= bn_bwd_apply(dy, y, bwd5, relu) bit for bit, without the pass)
        N, Cc, T, H, W = plan.src_shape
        nb = C.slv_cl16_stem_wgrad_ws_bytes(N, Cc, T, H, W, plan.Cout)
        ws = _ops.workspace(nb, dy.device)
        ay, ab5, arelu = bn_apply if bn_apply is not None else (None, None, False)
        C.slv_cl16_stem_wgrad(ptr(x_in.contiguous()), ptr(dy), ptr(dw), ptr(ws), nb, N, Cc, T, H, W, plan.Cout, ptr(ay),
                              ptr(ab5), int(arelu), stream())
        return dw
    assert bn_apply is None
    if plan.stem:
        x_in = patch if patch is not None else _patch(plan, x_in)
    if plan.chunks is not None:
        n = len(plan.chunks)
        slices = torch.empty(n, plan.Cout, n_w, dtype=torch.float32, device=dy.device)
        for i, (b0, b1, sub) in enumerate(plan.chunks):
            conv_wgrad(sub, dy[b0:b1], x_in[b0:b1], in_ss=in_ss, in_relu=in_relu, out=slices[i])
        C.slv_sum_slices(ptr(slices), ptr(dw), n, dw.numel(), stream())
        return dw
    ws = _ops.workspace(plan.ws_wgrad, dy.device)
    C.slv_cl16_wgrad(plan.g_wgrad.ctypes.data, plan.wm, plan.wn, ptr(dy), ptr(x_in), ptr(in_ss), ptr(dw), plan.Cout,
                     plan.patch_kw, ptr(ws), plan.ws_wgrad, stream())
    return dw


def _pc(x, C_real):
    """(positions, padded channels) of a channels-last tensor."""
    Cp = x.shape[-1]
    assert x.dtype == torch.bfloat16 and Cp == pad32(C_real)
    return x.numel() // Cp, Cp


def bn_act(x, ss, res=None, res_ss=None, relu=True, res_relu=False):
    """relu?(x*s + h + residual); residual = res, or res*rs + rh (res_ss), or -- res_relu -- the bf16-rounded relu(res*rs + rh)
    a consumer's load prologue would make of the raw tensor ``res``."""
    Cc = ss.shape[1]
    P, Cp = _pc(x, Cc)
    out = torch.empty_like(x)
    assert not res_relu or res_ss is not None
    C.slv_cl16_bn_act(ptr(x), ptr(ss), ptr(res), ptr(res_ss), int(relu) | (2 if res_relu else 0), ptr(out), P, Cc, Cp, stream())
    return out


def bn_bwd(g, x, mi, gamma, ss_mask=None, v_mask=None, x2=None, mi2=None, gamma2=None, ss2=None, sync=None,
           dgamma=None, dbeta=None, dgamma2=None, dbeta2=None, part=None):
    """ops.bn_bwd on channels-last bf16 tensors.  Returns (bwd5, bwd5_2, g_masked).  ``part``: the partial sums when
    the backward-data conv that produced g already formed them (conv_dgrad(bnr=...))."""
    Cc = gamma.numel()
    P, Cp = _pc(x, Cc)
    dev = x.device
    part2 = gout = None
    if part is not None:
        assert ss_mask is not None and v_mask is None and x2 is None
        ns = part.shape[1]
    else:
        ns = C.slv_cl16_bn_bwd_nsplit(P, Cp)
        part = torch.empty(Cc, ns, 2, dtype=torch.float32, device=dev)
        part2 = torch.empty(Cc, ns, 2, dtype=torch.float32, device=dev) if x2 is not None else None
        gout = torch.empty_like(g) if v_mask is not None else None
        C.slv_cl16_bn_bwd_reduce(ptr(g), ptr(x), ptr(mi), ptr(ss_mask), ptr(v_mask), ptr(x2), ptr(mi2), ptr(gout),
                                 ptr(part), ptr(part2), P, Cc, Cp, ns, stream())
    outs = _ops.bn_bwd_finish(part, part2, ns, float(P), mi, gamma, ss_mask, mi2, gamma2, sync, dgamma, dbeta, dgamma2,
                              dbeta2)
    return outs[0], outs[1], gout


def bn_bwd_apply(g, x, b5, relu, out=None):
    Cc = b5.shape[1]
    P, Cp = _pc(x, Cc)
    out = g if out is None else out
    C.slv_cl16_bn_bwd_apply(ptr(g), ptr(x), ptr(b5), int(relu), ptr(out), P, Cc, Cp, stream())
    return out


def avgpool_fwd(v, channels=None):
    N, Cp = v.shape[0], v.shape[-1]
    Cc = channels if channels is not None else Cp
    out = torch.empty(N, Cc, dtype=torch.float32, device=v.device)
    C.slv_avgpool_cl16(ptr(v), ptr(out), N, v.numel() // (N * Cp), Cc, Cp, stream())
    return out


def avgpool_bwd(dout, like):
    N, Cp = like.shape[0], like.shape[-1]
    dv = torch.empty_like(like)
    C.slv_cl16_avgpool_bwd(ptr(dout.contiguous()), ptr(dv), N, like.numel() // (N * Cp), dout.shape[1], Cp, stream())
    return dv


def from_channels_last16(x, channels):
    """bf16 [N,T,H,W,Cp] -> fp32 N,C,T,H,W (tests, inspection)."""
    N, T, H, W, Cp = x.shape
    y = torch.empty(N, channels, T, H, W, dtype=torch.float32, device=x.device)
    C.slv_from_cl16(ptr(x), ptr(y), N, channels, Cp, T * H * W, stream())
    return y
