"""Host side of the 16-bit MFMA path (first kernel: csrc/conv_cl16.hip).  Experimental, opt-in, forward only.

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
