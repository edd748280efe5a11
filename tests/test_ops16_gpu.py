"""16-bit MFMA path, first kernel (csrc/conv_cl16.hip): bf16 channels-last forward conv with fused affine + residual +
ReLU, against torch's fp32 conv on the SAME bf16-rounded inputs and weights (products of bf16 values are exact in
fp32, so only the accumulation order and the final rounding of the output to bf16 differ)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # N, Cin, T, H, W, Cout, k, stride, pad        every conv family of the two trunks, small extents
    (2, 64, 4, 12, 12, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # layer-1 spatial (M tile 9)
    (2, 144, 5, 10, 10, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),      # temporal, padded input channels (144 -> 160)
    (1, 64, 6, 14, 14, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),      # stride-2 spatial, Cout 230 -> 256
    (1, 230, 7, 7, 7, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),       # stride-2 temporal over an odd extent
    (2, 64, 4, 8, 8, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),        # downsample
    (2, 3, 3, 20, 20, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)),        # video stem
    (3, 1, 1, 40, 36, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),        # audio stem (2-D: T = 1)
    (2, 128, 1, 9, 7, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # audio BasicBlock conv
    (1, 512, 2, 7, 7, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # layer-4 spatial (many K-steps, 8 M blocks)
    (2, 128, 4, 14, 14, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # layer-2 spatial: 7 position tiles of the 8-wave kernel
    (1, 288, 6, 10, 10, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),     # layer-2 temporal: 27 chunks (odd)
    (1, 460, 7, 7, 7, 256, (3, 1, 1), (2, 1, 1), (1, 0, 0)),       # 460 -> 480 channels, stride-2 temporal
]


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("case", CASES)
def test_conv_cl16_matches_fp32_conv_on_bf16_values(case, g8):
    from selavi_amd import ops16
    N, Cin, T, H, W, Cout, k, st, pd = case
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = _bf(torch.randn(N, Cin, T, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, *k, generator=g) * (Cin * k[0] * k[1] * k[2]) ** -0.5)
    ss = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2])
    ref = F.conv3d(x.double(), w.double(), stride=st, padding=pd)
    res = _bf(torch.randn(ref.shape, generator=g))
    conv = ops16.Conv16(w.cuda(), st, pd)
    xc = ops16.to_channels_last16(x.cuda())
    assert xc.shape == (N, T, H, W, ops16.pad32(Cin)) and (xc[..., Cin:] == 0).all()
    assert torch.equal(xc[..., :Cin].float().cpu(), x.permute(0, 2, 3, 4, 1))          # layout conversion is exact here
    for use_ss, use_res, relu in ((False, False, False), (True, False, True), (True, True, True)):
        want = ref.clone()
        if use_ss:
            want = want * ss[0].double().view(1, -1, 1, 1, 1) + ss[1].double().view(1, -1, 1, 1, 1)
        if use_res:
            want = want + res.double()
        if relu:
            want = want.clamp_min(0)
        rc = ops16.to_channels_last16(res.cuda()) if use_res else None
        y = conv(xc, scale_shift=ss.cuda().contiguous() if use_ss else None, res=rc, relu=relu)
        assert y.shape[-1] == ops16.pad32(Cout) and (y[..., Cout:] == 0).all()               # padding channels are zero
        got = y[..., :Cout].float().cpu().permute(0, 4, 1, 2, 3).double()
        # fp32 accumulation of exact products + one rounding to bf16 (2^-9 relative)
        err = (got - want).abs()
        bound = want.abs() * 2.0 ** -8 + 1e-3 * want.abs().max()
        assert (err <= bound).all(), (case, float(err.max()), float(want.abs().max()))


def test_conv_cl16_rejects_bad_geometry():
    from selavi_amd import _lib, ops16
    conv = ops16.Conv16(torch.randn(16, 8, 1, 3, 3).cuda(), (1, 1, 1), (0, 1, 1))
    x = ops16.to_channels_last16(torch.randn(1, 8, 2, 6, 6).cuda())
    g = np.array([1, 2, 6, 6, 32, 16, 32, 2, 6, 7, 1, 3, 3, 1, 1, 1, 0, 1, 1, conv.Mrows], dtype=np.int32)   # Wo wrong
    with pytest.raises(_lib.SelaviHipError):
        _lib.C.slv_conv_cl16_fwd(g.ctypes.data, conv.mt, _lib.ptr(x), _lib.ptr(conv.wl), _lib.ptr(x), 0, 0, 0, _lib.stream())


@pytest.mark.parametrize("case", [(2, 3, 3, 20, 22, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
                                  (3, 1, 1, 40, 37, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3))])
def test_stem_conv_on_the_w_patch_layout(case):
    """StemConv16: the 7x7 stems (3 or 1 input channels) as a (1,7,1) conv over the 32-channel W-patch layout."""
    from selavi_amd import ops16
    N, Cin, T, H, W, Cout, k, st, pd = case
    g = torch.Generator().manual_seed(7)
    x = _bf(torch.randn(N, Cin, T, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, *k, generator=g) * 0.1)
    ss = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2])
    want = F.conv3d(x.double(), w.double(), stride=st, padding=pd)
    want = (want * ss[0].double().view(1, -1, 1, 1, 1) + ss[1].double().view(1, -1, 1, 1, 1)).clamp_min(0)
    y = ops16.StemConv16(w.cuda(), st, pd)(x.cuda(), scale_shift=ss.cuda().contiguous(), relu=True)
    got = y[..., :Cout].float().cpu().permute(0, 4, 1, 2, 3).double()
    assert got.shape == want.shape and (y[..., Cout:] == 0).all()
    assert ((got - want).abs() <= want.abs() * 2.0 ** -8 + 1e-3 * want.abs().max()).all()


def test_conv_cl16_random_geometries():
    """Property sweep: random shapes (ragged position counts, every M tile, channel counts that pad differently on the
    input and output side, strides 1/2, kernels up to 3x3x3) against torch's fp32 conv on the same bf16 values."""
    from selavi_amd import ops16
    rng = np.random.RandomState(20)
    n_done = 0
    while n_done < 24:
        k = tuple(int(v) for v in rng.choice([1, 3], size=3))
        st = tuple(int(v) for v in rng.choice([1, 2], size=3))
        pd = tuple(int(rng.randint(0, kk // 2 + 1)) for kk in k)
        N, T, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 6)), int(rng.randint(3, 15)), int(rng.randint(3, 15))
        if any((d + 2 * p - kk) // s + 1 <= 0 or d + 2 * p < kk for d, p, kk, s in zip((T, H, W), pd, k, st)):
            continue
        Cin = int(rng.choice([1, 3, 16, 45, 64, 144, 230]))
        Cout = int(rng.choice([8, 45, 64, 128, 144, 230, 300]))
        g = torch.Generator().manual_seed(n_done)
        x = _bf(torch.randn(N, Cin, T, H, W, generator=g))
        w = _bf(torch.randn(Cout, Cin, *k, generator=g) * (Cin * k[0] * k[1] * k[2]) ** -0.5)
        ss = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2])
        want = F.conv3d(x.double(), w.double(), stride=st, padding=pd)
        want = (want * ss[0].double().view(1, -1, 1, 1, 1) + ss[1].double().view(1, -1, 1, 1, 1)).clamp_min(0)
        y = ops16.Conv16(w.cuda(), st, pd)(ops16.to_channels_last16(x.cuda()), scale_shift=ss.cuda().contiguous(), relu=True)
        got = y[..., :Cout].float().cpu().permute(0, 4, 1, 2, 3).double()
        assert got.shape == want.shape, (k, st, pd, N, T, H, W, Cin, Cout)
        assert (y[..., Cout:] == 0).all()
        err = (got - want).abs()
        assert (err <= want.abs() * 2.0 ** -8 + 1e-3 * want.abs().max() + 1e-6).all(), (k, st, pd, N, T, H, W, Cin, Cout, float(err.max()))
        n_done += 1
