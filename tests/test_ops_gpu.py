"""GPU parity tests of the individual HIP ops (through the C ABI) against torch's CPU fp32 ops.
Tolerance: fp32 with a different summation order -> rtol 2e-4 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["x3", "native"], autouse=True)
def conv_arithmetic(request):
    """Every op test runs on both conv arithmetics of the fp32 path: "x3" (default; csrc/igemm3.hpp: operands split exactly
    into three bf16 pieces, six partial products on the bf16 matrix cores, fp32 accumulation) and "native" (csrc/igemm.hpp:
    v_mfma_f32_16x16x4_f32) -- same tolerances for both (5e-6 L2 against fp64 where the test has an fp64 reference)."""
    from selavi_amd import ops
    ops.set_conv_arithmetic(request.param)
    yield request.param
    ops.set_conv_arithmetic("x3")

# (Bn, Cin, T, H, W, Cout, k, stride, pad)  -- every conv family of R(2+1)D-18 / ResNet-9, odd channels,
# stride 2, odd T (T=15 -> 8), tails in M, N and K
GEOMS = [
    (2, 3, 4, 20, 20, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)),      # stem.0
    (2, 45, 5, 10, 10, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),     # stem.3
    (2, 64, 3, 12, 12, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # layer1 spatial
    (2, 144, 4, 6, 6, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),      # layer1 temporal
    (2, 64, 3, 12, 12, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),    # layer2 spatial stride 2
    (2, 230, 15, 6, 6, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),    # temporal stride 2, odd T
    (3, 64, 5, 9, 9, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),      # downsample 1x1x1 stride 2
    (1, 256, 2, 7, 7, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # layer4 spatial (921)
    (1, 921, 4, 4, 4, 512, (3, 1, 1), (2, 1, 1), (1, 0, 0)),     # layer4 temporal
    (2, 512, 2, 3, 3, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # layer4.1 spatial
    (3, 1, 1, 40, 36, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),      # audio conv1 (2-D)
    (3, 64, 1, 10, 9, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # audio 3x3 stride 2
    (3, 128, 1, 5, 5, 256, (1, 1, 1), (1, 2, 2), (0, 0, 0)),     # audio downsample
    (1, 5, 1, 3, 1, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1)),         # degenerate W = 1
    (2, 16, 1, 4, 4, 8, (3, 1, 1), (2, 1, 1), (1, 0, 0)),        # T = 1 under temporal stride 2 (empty parity class)
    (2, 8, 3, 5, 5, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0)),        # 1x1x1 stride 2 on odd extents
    (2, 256, 1, 2, 2, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # 8 output positions: a column tile with fewer valid
    (1, 512, 3, 1, 2, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),     #  lanes than taps (2 clips at 32x32 reach layer 4 so)
    (2, 64, 1, 3, 3, 96, (1, 3, 3), (1, 2, 2), (0, 1, 1)),       # same, strided (backward data by parity classes)
    (2, 16, 4, 6, 6, 40, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 27 taps (not in the trunks: the widest class the weight images take)
    (1, 32, 5, 7, 7, 24, (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # 27 taps, stride 2 everywhere: 8 parity classes of 1-8 taps
]


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _close_l2(got, want, rtol=5e-6):
    """L2-relative error against an fp64 reference: the fp32-MFMA GEMMs sit at ~6e-7 (tools/split_err.py)."""
    got = got.detach().cpu().double().flatten()
    want = want.detach().double().flatten()
    err = (got - want).norm().item() / (want.norm().item() + 1e-300)
    assert err <= rtol, f"L2-relative error {err:.3e} > {rtol:.1e}"


def _close(got, want, rtol=2e-4):
    got = got.detach().cpu().double()
    want = want.detach().double()
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("geo", GEOMS)
def test_conv_forward_dgrad_wgrad(geo):
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 1)
    w = _mk((Cout, Cin) + k, 2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
    xg, wg = x.to(dev), w.to(dev)

    # ---- plain forward + statistics partials
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, stride=st, padding=pd)
    y, ssum, ssq = ops.conv_fwd(plan, xg, wg)
    assert tuple(y.shape) == tuple(y_ref.shape)
    _close(y, y_ref)
    _close(ssum.sum(1), y_ref.sum((0, 2, 3, 4)), rtol=1e-3)
    _close(ssq.sum(1), (y_ref ** 2).sum((0, 2, 3, 4)), rtol=1e-3)

    # ---- plain dgrad / wgrad
    dy = _mk(tuple(y_ref.shape), 3)
    y_ref.backward(dy)
    dyg = dy.to(dev)
    wt = ops.conv_wt_transform(plan, wg)
    dx = ops.conv_dgrad(plan, dyg, wt)
    _close(dx, xr.grad)
    dw = ops.conv_wgrad(plan, dyg, xg)
    _close(dw.view_as(w), wr.grad)

    # ---- dgrad with addend (residual / in-place accumulate)
    add = _mk(tuple(x.shape), 4)
    dx2 = ops.conv_dgrad(plan, dyg, wt, addend=add.to(dev))
    _close(dx2, xr.grad + add)
    acc = add.to(dev).clone()
    ops.conv_dgrad(plan, dyg, wt, addend=acc, out=acc)
    _close(acc, xr.grad + add)


@pytest.mark.parametrize("geo", [GEOMS[4], GEOMS[7], GEOMS[8], GEOMS[9], GEOMS[14]])
def test_conv_every_launch_configuration(geo):
    """Every (tile, K-split) candidate the benchmark mode may pick (the cudnn.benchmark counterpart,
    main.py:187) computes the same convolution, in-place residual add and BN statistics."""
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 11)
    w = _mk((Cout, Cin) + k, 12, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)      # fp64 reference
    y_ref = F.conv3d(xr, wr, stride=st, padding=pd)
    dy = _mk(tuple(y_ref.shape), 13)
    y_ref.backward(dy.double())
    add = _mk(tuple(x.shape), 14)
    plan = ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)     # private plan: configs are mutated
    xg, wg, dyg = x.to(dev), w.to(dev), dy.to(dev)
    wt = ops.conv_wt_transform(plan, wg)
    ncfg = 0
    for op in range(3):
        cands = plan.candidates(op)
        assert cands, "no launch configuration offered"
        for cfg in cands:
            cfgs = [0, 0, 0]
            cfgs[op] = cfg
            plan.set_configs(*cfgs)
            ncfg += 1
            if op == 0:
                y, ssum, ssq = ops.conv_fwd(plan, xg, wg)
                _close_l2(y, y_ref)
                _close(ssum.sum(1), y_ref.sum((0, 2, 3, 4)), rtol=1e-3)
                _close_l2(ssq.sum(1), (y_ref ** 2).sum((0, 2, 3, 4)), rtol=1e-5)
                y2, _, _ = ops.conv_fwd(plan, xg, wg, want_stats=False)
                assert torch.equal(y, y2)
            elif op == 1:
                acc = add.to(dev).clone()
                ops.conv_dgrad(plan, dyg, wt, addend=acc, out=acc)
                _close_l2(acc, xr.grad + add.double())
                # fused BatchNorm-backward reduction of the producing layer (epilogue / split-K reduce)
                xp = _mk(tuple(x.shape), 15)
                ss = torch.stack([_mk((Cin,), 16).abs() + 0.5, _mk((Cin,), 17) * 0.3])
                mi = torch.stack([_mk((Cin,), 18) * 0.2, _mk((Cin,), 19).abs() + 0.5])
                acc2 = add.to(dev).clone()
                dx2, part = ops.conv_dgrad(plan, dyg, wt, addend=acc2, out=acc2,
                                           bnr=(xp.to(dev), ss.to(dev).contiguous(), mi.to(dev).contiguous()))
                assert torch.equal(dx2, acc)
                v = lambda t: t.double().view(1, -1, 1, 1, 1)
                gfin = xr.grad + add.double()
                gm = gfin * ((xp.double() * v(ss[0]) + v(ss[1])) > 0)
                want0 = gm.sum((0, 2, 3, 4))
                want1 = (gm * (xp.double() - v(mi[0])) * v(mi[1])).sum((0, 2, 3, 4))
                assert part.shape[0] == Cin and part.shape[2] == 2
                _close(part[:, :, 0].sum(1), want0, rtol=2e-4)
                _close(part[:, :, 1].sum(1), want1, rtol=2e-4)
            else:
                _close_l2(ops.conv_wgrad(plan, dyg, xg).view_as(w), wr.grad)
    assert ncfg >= 3
    with pytest.raises(Exception):
        plan.set_configs(3 | (2 << 8) | (1 << 16), 0, 0)        # 48-row tile does not exist
        ops.conv_fwd(plan, xg, wg)


@pytest.mark.parametrize("geo", [GEOMS[2], GEOMS[4], GEOMS[5], GEOMS[8], GEOMS[11]])
def test_conv_fused_bn_prologues(geo):
    """Consumer-side BN+ReLU on load (forward / wgrad B operand) and BN-backward on load
    (dgrad B operand / wgrad A operand) against the unfused torch composition."""
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 1)
    w = _mk((Cout, Cin) + k, 2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    s_in = _mk((Cin,), 5).abs() + 0.5
    h_in = _mk((Cin,), 6) * 0.3
    plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
    xa = torch.relu(x * s_in.view(1, -1, 1, 1, 1) + h_in.view(1, -1, 1, 1, 1)).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv3d(xa, wr, stride=st, padding=pd)
    ss_in = torch.stack([s_in, h_in]).to(dev).contiguous()
    y, _, _ = ops.conv_fwd(plan, x.to(dev), w.to(dev), in_ss=ss_in, in_relu=True)
    _close(y, y_ref)

    # gradient wrt raw conv output formed on load: dXout = A1*mask*g + A2 + A3*xo
    g = _mk(tuple(y_ref.shape), 7)
    xo = y_ref.detach()
    p5 = torch.stack([_mk((Cout,), 8).abs() + 0.2, _mk((Cout,), 9) * 0.2, _mk((Cout,), 10), _mk((Cout,), 11) * 0.1,
                      _mk((Cout,), 12) * 0.1])
    v = lambda i: p5[i].view(1, -1, 1, 1, 1)
    for relu in (True, False):
        mask = ((xo * v(0) + v(1)) > 0).float() if relu else 1.0
        dxo = v(2) * mask * g + v(3) + v(4) * xo
        gx, gw = torch.autograd.grad(y_ref, (xa, wr), dxo, retain_graph=True)
        wt = ops.conv_wt_transform(plan, w.to(dev))
        dx = ops.conv_dgrad(plan, g.to(dev), wt, x_out=xo.to(dev), bwd5=p5.to(dev).contiguous(), relu=relu)
        _close(dx, gx)
        dw = ops.conv_wgrad(plan, g.to(dev), x.to(dev), x_out=xo.to(dev), bwd5=p5.to(dev).contiguous(), a_relu=relu,
                            in_ss=ss_in, in_relu=True)
        _close(dw.view_as(w), gw)


@pytest.mark.parametrize("M,N,K", [(16, 309, 512), (1000, 512, 512), (777, 28, 512), (130, 400, 96)])
def test_gemm_nt(M, N, K):
    from selavi_amd import ops
    A, B, bias = _mk((M, K), 1), _mk((N, K), 2, K ** -0.5), _mk((N,), 3)
    out = ops.gemm_nt(A.cuda(), B.cuda(), bias.cuda())
    _close(out, A @ B.t() + bias)
    out = ops.gemm_nt(A.cuda(), B.cuda())
    _close(out, A @ B.t())


def test_bn_train_chain_matches_torch():
    """conv -> BN(train) -> ReLU -> conv -> BN(train) + residual -> ReLU, forward and backward,
    with every BN fused as the engine does it."""
    from selavi_amd import ops
    dev = torch.device("cuda")
    Bn, Cc, T, H, W = 3, 24, 3, 6, 5
    x = _mk((Bn, Cc, T, H, W), 1)
    w1 = _mk((40, Cc, 1, 3, 3), 2, 0.1)
    w2 = _mk((Cc, 40, 3, 1, 1), 3, 0.1)
    g1, b1 = _mk((40,), 4).abs() + 0.5, _mk((40,), 5) * 0.2
    g2, b2 = _mk((Cc,), 6).abs() + 0.5, _mk((Cc,), 7) * 0.2
    # torch reference
    P = [t.clone().requires_grad_(True) for t in (x, w1, w2, g1, b1, g2, b2)]
    xr, w1r, w2r, g1r, b1r, g2r, b2r = P
    rm1, rv1, rm2, rv2 = torch.zeros(40), torch.ones(40), torch.zeros(Cc), torch.ones(Cc)
    y1 = F.conv3d(xr, w1r, padding=(0, 1, 1))
    a1 = F.relu(F.batch_norm(y1, rm1, rv1, g1r, b1r, True, 0.1, 1e-5))
    y2 = F.conv3d(a1, w2r, padding=(1, 0, 0))
    out = F.relu(F.batch_norm(y2, rm2, rv2, g2r, b2r, True, 0.1, 1e-5) + xr)
    dout = _mk(tuple(out.shape), 8)
    out.backward(dout)
    # HIP
    p1 = ops.ConvPlan.get((Bn, Cc, T, H, W), 40, (1, 3, 3), (1, 1, 1), (0, 1, 1), dev)
    p2 = ops.ConvPlan.get((Bn, 40, T, H, W), Cc, (3, 1, 1), (1, 1, 1), (1, 0, 0), dev)
    xg = x.to(dev)
    G = [t.to(dev) for t in (w1, w2, g1, b1, g2, b2)]
    w1g, w2g, g1g, b1g, g2g, b2g = G
    R = [torch.zeros(40, device=dev), torch.ones(40, device=dev), torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)]
    Y1, s1, q1 = ops.conv_fwd(p1, xg, w1g)
    mi1, ss1 = ops.bn_train_finalize(s1, q1, p1.count, g1g, b1g, R[0], R[1], 0.1, 1e-5)
    Y2, s2, q2 = ops.conv_fwd(p2, Y1, w2g, in_ss=ss1, in_relu=True)
    mi2, ss2 = ops.bn_train_finalize(s2, q2, p2.count, g2g, b2g, R[2], R[3], 0.1, 1e-5)
    V = ops.bn_act(Y2, ss2, res=xg, relu=True)
    _close(V, out)
    _close(R[0], rm1, 1e-3), _close(R[1], rv1, 1e-3), _close(R[3], rv2, 1e-3)
    # backward
    dg2, db2 = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    b5_2, _, dz = ops.bn_bwd(dout.to(dev), Y2, mi2, g2g, v_mask=V, dgamma=dg2, dbeta=db2)
    _close(dg2, g2r.grad, 1e-3), _close(db2, b2r.grad, 1e-3)
    dw2 = ops.conv_wgrad(p2, dz, Y1, x_out=Y2, bwd5=b5_2, a_relu=False, in_ss=ss1, in_relu=True)
    _close(dw2.view_as(w2), w2r.grad, 1e-3)
    da1 = ops.conv_dgrad(p2, dz, ops.conv_wt_transform(p2, w2g), x_out=Y2, bwd5=b5_2, relu=False)
    dg1, db1 = torch.empty(40, device=dev), torch.empty(40, device=dev)
    b5_1, _, _ = ops.bn_bwd(da1, Y1, mi1, g1g, ss_mask=ss1, dgamma=dg1, dbeta=db1)
    _close(dg1, g1r.grad, 1e-3), _close(db1, b1r.grad, 1e-3)
    dw1 = ops.conv_wgrad(p1, da1, xg, x_out=Y1, bwd5=b5_1, a_relu=True)
    _close(dw1.view_as(w1), w1r.grad, 1e-3)
    dx = ops.conv_dgrad(p1, da1, ops.conv_wt_transform(p1, w1g), x_out=Y1, bwd5=b5_1, relu=True, addend=dz)
    _close(dx, xr.grad, 1e-3)


def test_pools_and_sgd():
    from selavi_amd import ops
    dev = torch.device("cuda")
    v = _mk((3, 10, 2, 7, 7), 1)
    _close(ops.avgpool_fwd(v.to(dev)), v.mean((2, 3, 4)))
    d = _mk((3, 10), 2)
    _close(ops.avgpool_bwd(d.to(dev), v.to(dev)), (d / 98).view(3, 10, 1, 1, 1).expand_as(v))
    # relu(bn(x)) -> maxpool(3,2,1)
    x = _mk((2, 6, 1, 11, 9), 3).requires_grad_(True)
    s, h = _mk((6,), 4).abs() + 0.5, _mk((6,), 5) * 0.3
    a = F.relu(x * s.view(1, -1, 1, 1, 1) + h.view(1, -1, 1, 1, 1))
    pooled = F.max_pool2d(a[:, :, 0], 3, 2, 1)
    out, idx = ops.bnrelu_maxpool_fwd(x.detach().to(dev), torch.stack([s, h]).to(dev))
    _close(out[:, :, 0], pooled)
    dp = _mk(tuple(pooled.shape), 6)
    ga, = torch.autograd.grad(pooled, a, dp)
    dy = ops.maxpool_bwd(dp.unsqueeze(2).contiguous().to(dev), idx, tuple(x.shape))
    # positions where relu output is 0 may tie; their gradient is masked by the ReLU backward anyway
    m = (a > 0).float()
    _close(dy.cpu() * m, ga * m)
    # SGD
    ps = [_mk((5000,), 7), _mk((33, 7), 8), _mk((1,), 9)]
    gs = [_mk(tuple(p.shape), 10 + i) for i, p in enumerate(ps)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.SGD(ref, lr=0.01, momentum=0.9, weight_decay=1e-5)
    pg = [p.to(dev) for p in ps]
    bufs = [torch.zeros_like(p) for p in pg]
    for step in range(3):
        for r, g in zip(ref, gs):
            r.grad = g * (step + 1)
        opt.step()
        ops.sgd_step(pg, [(g * (step + 1)).to(dev) for g in gs], bufs, 0.01, 0.9, 1e-5, step == 0)
    for p, r in zip(pg, ref):
        _close(p, r.detach(), 1e-6)


# ------------------------------------------------------------------ full cfg2 sizes (B = 16)
FULL = [
    ("layer1 spatial", (16, 64, 16, 56, 56, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))),
    ("layer2.0 spatial stride 2", (16, 64, 16, 56, 56, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1))),
    ("layer2.0 temporal stride 2", (16, 230, 16, 28, 28, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0))),
    ("layer4.1 spatial", (16, 512, 2, 7, 7, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1))),
    ("stem.0", (16, 3, 16, 112, 112, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3))),
]


@pytest.mark.parametrize("name,geo", FULL, ids=[n for n, _ in FULL])
@pytest.mark.parametrize("tuned", [False, True], ids=["heuristic", "benchmark-mode"])
def test_full_size_adjoint_identities(name, geo, tuned):
    """BASELINE cfg2 sizes, where no CPU oracle is affordable: size-independent identities.
    conv is bilinear, so for random x, w, g:  <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(g, x)>
    (fp64 inner products on the device), and the fused BatchNorm statistics of the forward epilogue
    equal the sums of the tensor it wrote.  The forward itself is pinned at small sizes against fp64
    (above) and here by linearity: conv(x1 + 2*x2) = conv(x1) + 2*conv(x2)."""
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(Bn, Cin, T, H, W, device=dev, generator=gen)
    x2 = torch.randn(Bn, Cin, T, H, W, device=dev, generator=gen)
    w = torch.randn(Cout, Cin, *k, device=dev, generator=gen) * (Cin * k[0] * k[1] * k[2]) ** -0.5
    was = ops.benchmark
    ops.benchmark = tuned
    try:
        plan = ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)     # private plan (tuned or heuristic)
    finally:
        ops.benchmark = was
    wf, wt = ops.conv_w_transform(plan, w)
    y, ssum, ssq = ops.conv_fwd(plan, x, w, wf=wf)
    g = torch.randn(y.shape, device=dev, generator=gen)
    dx = ops.conv_dgrad(plan, g, wt)
    dw = ops.conv_wgrad(plan, g, x).view_as(w)
    dot = lambda a, b: float((a.double().flatten() * b.double().flatten()).sum())
    a, b, c = dot(y, g), dot(x, dx), dot(w, dw)
    scale = (dot(y, y) * dot(g, g)) ** 0.5
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    # statistics epilogue vs the stored tensor
    np.testing.assert_allclose(ssum.double().sum(1).cpu().numpy(), y.double().sum((0, 2, 3, 4)).cpu().numpy(),
                               rtol=1e-4, atol=1e-3 * float(y.abs().max()) * 10)
    np.testing.assert_allclose(ssq.double().sum(1).cpu().numpy(), (y.double() ** 2).sum((0, 2, 3, 4)).cpu().numpy(),
                               rtol=1e-5)
    # linearity of the forward
    y12, _, _ = ops.conv_fwd(plan, x + 2 * x2, w, wf=wf, want_stats=False)
    y2, _, _ = ops.conv_fwd(plan, x2, w, wf=wf, want_stats=False)
    err = float(((y12.double() - (y.double() + 2 * y2.double())).norm()) / y12.double().norm())
    assert err <= 2e-6, err


@pytest.mark.parametrize("geo", [(7, 20, 4, 12, 12, 24, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                 (5, 16, 6, 10, 10, 40, (3, 1, 1), (2, 1, 1), (1, 0, 0)),
                                 (6, 8, 4, 12, 12, 16, (1, 1, 1), (2, 2, 2), (0, 0, 0))])
def test_batch_sliced_conv_equals_unsliced(geo, monkeypatch):
    """Tensors beyond the 32-bit buffer range (BASELINE configs[4]: 128 clips x 32 frames per GPU) are convolved
    in batch slices (ops.ConvPlan.chunks).  Forced here on small tensors: clips are independent, so the forward and
    the data gradient are bit-identical to the unsliced launch; the statistics and the weight gradient differ only
    in summation order."""
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x, w = _mk((Bn, Cin, T, H, W), 21).to(dev), _mk((Cout, Cin) + k, 22, scale=0.1).to(dev)
    ss = torch.stack([_mk((Cin,), 23).abs() + 0.5, _mk((Cin,), 24) * 0.3]).to(dev).contiguous()
    mi = torch.stack([_mk((Cin,), 25) * 0.2, _mk((Cin,), 26).abs() + 0.5]).to(dev).contiguous()
    whole = ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)
    assert whole.chunks is None
    per_clip = 4 * max(x[0].numel(), whole.out_shape[1] * whole.P_out)
    monkeypatch.setattr(ops, "CONV_BUF_LIMIT", 2 * per_clip + 64)             # at most two clips per slice
    sliced = ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)
    assert sliced.chunks is not None and len(sliced.chunks) == -(-Bn // 2)
    assert [c[1] - c[0] for c in sliced.chunks] == sorted([c[1] - c[0] for c in sliced.chunks], reverse=True)
    dy, add, xp = _mk(whole.out_shape, 27).to(dev), _mk(tuple(x.shape), 28).to(dev), _mk(tuple(x.shape), 29).to(dev)
    wf, wt = ops.conv_w_transform(whole, w)
    y0, s0, q0 = ops.conv_fwd(whole, x, w, in_ss=ss, in_relu=True, wf=wf)
    y1, s1, q1 = ops.conv_fwd(sliced, x, w, in_ss=ss, in_relu=True, wf=wf)
    assert torch.equal(y0, y1) and s1.shape[0] == Cout
    _close(s1.double().sum(1), s0.double().sum(1).cpu(), rtol=1e-5)
    _close(q1.double().sum(1), q0.double().sum(1).cpu(), rtol=1e-6)
    (d0, p0), (d1, p1) = (ops.conv_dgrad(pl, dy, wt, addend=add, bnr=(xp, ss, mi)) for pl in (whole, sliced))
    assert torch.equal(d0, d1)
    _close(p1.double().sum(1), p0.double().sum(1).cpu(), rtol=1e-5)
    assert torch.equal(ops.conv_dgrad(sliced, dy, wt), ops.conv_dgrad(whole, dy, wt))
    g0 = ops.conv_wgrad(whole, dy, x, in_ss=ss, in_relu=True)
    g1 = ops.conv_wgrad(sliced, dy, x, in_ss=ss, in_relu=True)
    _close_l2(g1, g0.double().cpu(), rtol=2e-6)
    with pytest.raises(ValueError):
        sliced.set_configs(0, 0, 0)
    monkeypatch.setattr(ops, "CONV_BUF_LIMIT", per_clip // 2)
    with pytest.raises(ValueError):
        ops.ConvPlan(Bn, Cin, T, H, W, Cout, k, st, pd, dev)


@pytest.mark.parametrize("geo", [GEOMS[2], GEOMS[3], GEOMS[4], GEOMS[7], GEOMS[9]])
def test_split_operand_arithmetic_is_as_accurate_as_the_native_fp32_mfma(geo, conv_arithmetic):
    """The claim behind the default arithmetic (csrc/igemm3.hpp header): cutting both fp32 operands exactly into three bf16
    pieces and dropping the three smallest of the nine partial products leaves an error of the size of ONE fp32 rounding
    per product -- forward, backward data and weight gradient against fp64, side by side with the native fp32-input MFMA
    kernels on the same data: the split path's L2 error must not exceed 1.5 x the native path's (measured: 0.6-1.0 x; both
    ~3-6e-7), and on operands that ARE bf16 values (pieces 2 and 3 zero) it is as exact as fp32 accumulation allows."""
    if conv_arithmetic != "x3":
        pytest.skip("compares both arithmetics in one run")
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 11)
    w = _mk((Cout, Cin) + k, 12, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv3d(xr, wr, stride=st, padding=pd)
    dy = _mk(tuple(y64.shape), 13)
    y64.backward(dy.double())
    errs = {}
    for mode in ("native", "x3"):
        ops.set_conv_arithmetic(mode)
        plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
        y, _, _ = ops.conv_fwd(plan, x.to(dev), w.to(dev))
        dx = ops.conv_dgrad(plan, dy.to(dev), ops.conv_wt_transform(plan, w.to(dev)))
        dw = ops.conv_wgrad(plan, dy.to(dev), x.to(dev))
        rel = lambda got, want: float((got.detach().cpu().double().flatten() - want.flatten()).norm() / want.norm())
        errs[mode] = (rel(y, y64.detach()), rel(dx, xr.grad), rel(dw.view_as(w), wr.grad))
    ops.set_conv_arithmetic("x3")
    print("L2 error vs fp64 (fwd, dgrad, wgrad): native %s  x3 %s" % (["%.2e" % e for e in errs["native"]], ["%.2e" % e for e in errs["x3"]]))
    for en, e3 in zip(errs["native"], errs["x3"]):
        assert e3 <= 5e-6 and e3 <= 1.5 * en + 1e-7, (errs["native"], errs["x3"])
    # bf16-valued operands: the split is (x, 0, 0) and every product exact -> the x3 result equals fp32 accumulation of exact
    # products, i.e. as close to fp64 as the native kernel on the same values
    xb, wb = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    yb64 = F.conv3d(xb.double(), wb.double(), stride=st, padding=pd)
    plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
    yb, _, _ = ops.conv_fwd(plan, xb.to(dev), wb.to(dev))
    _close_l2(yb, yb64, rtol=2e-6)


def _conv64(x, w, dy, st, pd):
    """fp64 forward / backward data / weight gradient + the componentwise error scales sum |a||b| of each."""
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv3d(xr, wr, stride=st, padding=pd)
    y.backward(dy.double())
    xa, wa = x.double().abs().requires_grad_(True), w.double().abs().requires_grad_(True)
    ya = F.conv3d(xa, wa, stride=st, padding=pd)
    ya.backward(dy.double().abs())
    return (y.detach(), xr.grad, wr.grad), (ya.detach(), xa.grad, wa.grad)


RANGE_CASES = {                    # (x scale, w scale, dy scale)
    "tiny_activations": (1e-30, 1.0, 1.0),
    "huge_activations": (1e+30, 1.0, 1e-6),
    "tiny_weights_huge_gradient": (1.0, 1e-30, 1e+30),
    "huge_weights_tiny_gradient": (1e-6, 1e+30, 1e-30),
}


@pytest.mark.parametrize("case", list(RANGE_CASES) + ["twelve_decades"])
@pytest.mark.parametrize("geo", [GEOMS[2], GEOMS[3], GEOMS[4]])
def test_split_operand_arithmetic_over_the_fp32_range(geo, case, conv_arithmetic):
    """The three-piece split is exact wherever the THIRD piece (2^-16 of the value) is still a normal bf16 number and the
    first is finite: operands scaled to 1e-30 / 1e+30, and one tensor spanning twelve decades (late-training gradients: a
    few large entries among many tiny ones).  Held to the native fp32-input MFMA on the same data, against fp64 -- in L2 AND
    componentwise against sum |a||b| (the bound that says the small outputs of a wide-range tensor are right too)."""
    if conv_arithmetic != "x3":
        pytest.skip("compares both arithmetics in one run")
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = geo
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 21)
    w = _mk((Cout, Cin) + k, 22, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    To, Ho, Wo = ((T + 2 * pd[0] - k[0]) // st[0] + 1, (H + 2 * pd[1] - k[1]) // st[1] + 1, (W + 2 * pd[2] - k[2]) // st[2] + 1)
    dy = _mk((Bn, Cout, To, Ho, Wo), 23)
    if case == "twelve_decades":
        g = torch.Generator().manual_seed(24)
        x = x * 10.0 ** (torch.rand(x.shape, generator=g) * 12 - 6)
        dy = dy * 10.0 ** (torch.rand(dy.shape, generator=g) * 12 - 6)
        w = w * 10.0 ** (torch.rand(w.shape, generator=g) * 6 - 3)
    else:
        sx, sw, sd = RANGE_CASES[case]
        x, w, dy = x * sx, w * sw, dy * sd
    ref, scale = _conv64(x, w, dy, st, pd)
    errs = {}
    for mode in ("native", "x3"):
        ops.set_conv_arithmetic(mode)
        plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
        y, _, _ = ops.conv_fwd(plan, x.to(dev), w.to(dev))
        dx = ops.conv_dgrad(plan, dy.to(dev), ops.conv_wt_transform(plan, w.to(dev)))
        dw = ops.conv_wgrad(plan, dy.to(dev), x.to(dev)).view_as(w)
        e = []
        for got, want, sc in zip((y, dx, dw), ref, scale):
            got = got.detach().cpu().double()
            assert torch.isfinite(got).all()
            e.append((float((got - want).norm() / want.norm()), float(((got - want).abs() / (sc + 1e-300)).max())))
        errs[mode] = e
    ops.set_conv_arithmetic("x3")
    print(case, "(L2, componentwise) fwd/dgrad/wgrad: native", [("%.1e" % a, "%.1e" % b) for a, b in errs["native"]],
          " x3", [("%.1e" % a, "%.1e" % b) for a, b in errs["x3"]])
    for (l2n, cwn), (l23, cw3) in zip(errs["native"], errs["x3"]):
        assert l23 <= 5e-6 and l23 <= 1.5 * l2n + 1e-7, errs
        # componentwise: fp32 accumulation of n terms stays below ~sqrt(n) eps of sum |a||b|; the split adds <= 3 * 2^-24
        assert cw3 <= 2e-6 and cw3 <= 1.5 * cwn + 2e-7, errs


def test_non_finite_inputs_poison_the_same_outputs_on_both_arithmetics(conv_arithmetic):
    """+-inf / NaN activations: the native MFMA propagates them (inf or NaN); the split turns an infinity into (inf, NaN, NaN)
    -- a different non-finite VALUE, the same non-finite SET: every output whose receptive field holds the element is
    non-finite on both arithmetics, every other output is bit-identical to the clean run."""
    if conv_arithmetic != "x3":
        pytest.skip("compares both arithmetics in one run")
    from selavi_amd import ops
    Bn, Cin, T, H, W, Cout, k, st, pd = GEOMS[2]
    dev = torch.device("cuda")
    x = _mk((Bn, Cin, T, H, W), 31)
    w = _mk((Cout, Cin) + k, 32, scale=(Cin * 9) ** -0.5)
    bad = x.clone()
    bad[0, 5, 1, 4, 7] = float("inf")
    bad[1, 60, 2, 0, 0] = float("-inf")
    bad[1, 3, 0, 11, 11] = float("nan")
    hit = torch.zeros(Bn, 1, T, H, W)
    hit[0, 0, 1, 4, 7] = hit[1, 0, 2, 0, 0] = hit[1, 0, 0, 11, 11] = 1
    hit = F.conv3d(hit, torch.ones(1, 1, *k), stride=st, padding=pd)[:, 0] > 0          # outputs that see a poisoned input
    sets = {}
    for mode in ("native", "x3"):
        ops.set_conv_arithmetic(mode)
        plan = ops.ConvPlan.get((Bn, Cin, T, H, W), Cout, k, st, pd, dev)
        y_clean, _, _ = ops.conv_fwd(plan, x.to(dev), w.to(dev))
        y_bad, _, _ = ops.conv_fwd(plan, bad.to(dev), w.to(dev))
        nonfinite = ~torch.isfinite(y_bad).cpu()
        assert (nonfinite == hit[:, None].expand_as(nonfinite)).all(), mode
        assert torch.equal(y_bad.cpu()[~nonfinite], y_clean.cpu()[~nonfinite]), mode
        sets[mode] = nonfinite
    ops.set_conv_arithmetic("x3")
    assert torch.equal(sets["native"], sets["x3"])
