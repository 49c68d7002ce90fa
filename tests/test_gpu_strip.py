"""The strip kernel (cdl_strip.hip, reached through the cdl_fusedg_* entry points): the fused iteration for one image
channel, stride 1 or 2, up to 192 subbands -- the shipped CDLNet-s2030 architecture (K=30 M=169 P=7 s=2,
/root/reference/trained_nets/CDLNet-s2030/args.json:2-9) -- against the shape-generic kernels, stage by stage and as
whole sweeps.  (Net-level parity against the CPU oracle: tests/test_gpu_nets.py 's2030-arch', tests/test_gpu_depth.py.)"""
import pytest
import torch

from gpu_util import check, log

pytestmark = pytest.mark.gpu

# N, M, P, s, (H, W), masked
SHAPES = [
    (2, 169, 7, 2, (64, 128), False),       # the shipped filter bank: 6 channel tiles (the last one 9 real channels)
    (1, 169, 7, 2, (150, 154), True),       # Hz = 75, Wz = 77: ragged last segment and last strip
    (1, 64, 5, 2, (40, 72), False),         # P = 5 (one tap tile), two whole channel tiles
    (2, 24, 3, 2, (20, 36), False),         # P = 3, one partial channel tile
    (1, 100, 7, 1, (33, 70), True),         # unit stride with M > 64: 4 tiles (the last one 4 real channels)
    (1, 72, 5, 1, (18, 40), False),         # unit stride, P = 5, M not a multiple of 16
    (3, 169, 7, 2, (8, 8), False),          # code plane smaller than a strip / a segment
]


def make_geom(N, M, P, s, sp):
    import cdlnet_video_amd as cva
    return cva.ops.Geometry.make(N, 1, M, sp, (P, P), (P // 2, P // 2), s)


@pytest.mark.parametrize("N,M,P,s,sp,masked", SHAPES)
def test_strip_iteration_vs_generic(N, M, P, s, sp, masked):
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(sum(sp) + M)
    g = make_geom(N, M, P, s, sp)
    assert o.fusedg_supported(g) and not o.fused_supported(g)
    ish, csh, fsh = g.image_shape(), g.code_shape(), g.filter_shape()
    r = torch.randn(ish, generator=gen).cuda()
    z = (torch.randn(csh, generator=gen) * (torch.rand(csh, generator=gen) < 0.3)).cuda()
    wA = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    wB = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    tau = (torch.rand(N, M, generator=gen) * 0.6 + 0.01).cuda()
    yp = torch.randn(ish, generator=gen).cuda()
    mask = (torch.rand(ish, generator=gen) < 0.5).float().cuda() if masked else None
    frags = o.fusedg_prep(g, wA, wB)
    patches = o.fusedg_patches(g, "cuda")
    tag = f"strip N{N}M{M}P{P}s{s} {sp}"
    for name, zin, sgn in (("iter", z, -1.0), ("first", None, 1.0)):
        z_ref = o.analysis(g, r, wA, sgn, zin, None, tau)
        r_ref = o.synthesis(g, z_ref, wB, 1.0, None, mask, yp)
        patches.fill_(float("nan"))                       # every patch word that is read must be written
        bits = o.fusedg_map(g, "cuda").fill_(-1)          # every map word must be written
        z_got = o.fusedg_iter(g, r, zin, tau, frags, sgn, patches, map_out=bits)
        assert torch.equal(bits, o.fusedg_support_map(g, z_got)), f"{tag} {name}: support/sign map"
        r_got = o.fusedg_assemble(g, patches, mask, yp)
        check(f"{tag} {name} z'", z_got, z_ref, 2e-5)
        check(f"{tag} {name} r_next", r_got, o.synthesis(g, z_got, wB, 1.0, None, mask, yp), 2e-5)
        check(f"{tag} {name} r_next(end-to-end)", r_got, r_ref, 8e-5)
        assert float(((z_got != 0) == (z_ref != 0)).float().mean()) > 0.9999
        z2 = o.fusedg_iter(g, r, zin, tau, frags, sgn, patches)          # deterministic, with or without the map
        assert torch.equal(z2, z_got)
        assert torch.equal(o.fusedg_assemble(g, patches, mask, yp), r_got)


def test_strip_negative_and_nan_thresholds():
    """A negative threshold takes the general shrinkage (sign(u) relu(|u| - t), net.py:11-14), a NaN one yields NaN."""
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(3)
    N, M, P, s, sp = 2, 169, 7, 2, (32, 64)
    g = make_geom(N, M, P, s, sp)
    r = torch.randn(g.image_shape(), generator=gen).cuda()
    z = torch.randn(g.code_shape(), generator=gen).cuda()
    w = (torch.randn(g.filter_shape(), generator=gen) * 0.1).cuda()
    tau = (torch.rand(N, M, generator=gen) * 0.4 - 0.1).cuda()             # a quarter of them negative
    frags, patches = o.fusedg_prep(g, w, w), o.fusedg_patches(g, "cuda")
    got = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches)
    check("strip negative thresholds", got, o.analysis(g, r, w, -1.0, z, None, tau), 2e-5)
    tau[1, 168] = float("nan")
    got = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches)
    assert torch.isnan(got[1, 168]).all() and not torch.isnan(got[0]).any() and not torch.isnan(got[1, :168]).any()


@pytest.mark.parametrize("N,M,P,s,sp,masked", SHAPES[:5])
def test_strip_backward_stage_vs_generic(N, M, P, s, sp, masked):
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(7 * sum(sp) + M)
    g = make_geom(N, M, P, s, sp)
    ish, csh, fsh = g.image_shape(), g.code_shape(), g.filter_shape()
    thin = torch.randn(ish, generator=gen).cuda()
    base = torch.randn(csh, generator=gen).cuda()
    gate = (torch.randn(csh, generator=gen) * (torch.rand(csh, generator=gen) < 0.3)).cuda()
    w1 = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    w2 = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    c = torch.rand(N, generator=gen).cuda()
    mask = (torch.rand(ish, generator=gen) < 0.5).float().cuda() if masked else None
    frags = o.fusedg_prep(g, w1, w2)
    patches = o.fusedg_patches(g, "cuda")
    items = o._fusedg_sizes(g)[2]
    dtp = torch.empty(items, M, device="cuda")
    bits = o.fusedg_support_map(g, gate)
    tag = f"strip-bwd N{N}M{M}P{P}s{s} {sp}"
    for name, b in (("with-base", base), ("no-base", None)):
        gk = o.analysis(g, thin, w1, 1.0, b, None, None)
        du_ref = gk * (gate != 0)
        dt_ref = torch.zeros(2, M, device="cuda")
        o.tau_grad(g, gk, gate, c, dt_ref)
        q_ref = o.synthesis(g, gk, w2, -1.0, gate, mask, None)
        patches.fill_(float("nan"))
        dtp.fill_(float("nan"))
        du = o.fusedg_stage_bwd(g, thin, b, bits, frags, patches, dtp, True)
        q = o.fusedg_assemble(g, patches, mask, None, -1.0)
        dt = torch.zeros(2, M, device="cuda")
        o.fusedg_dtau_reduce(g, dtp, c, dt)
        check(f"{tag} {name} du", du, du_ref, 2e-5)
        # same support; the one exception allowed: an entry the two arithmetics cancel differently (|du| ~ 1e-6 of the max)
        differ = (du == 0) != (du_ref == 0)
        assert int(differ.sum()) <= 2 and float((du - du_ref)[differ].abs().max() if differ.any() else 0.0) < 1e-4
        check(f"{tag} {name} q", q, q_ref, 8e-5)
        check(f"{tag} {name} dt", dt, dt_ref, 8e-5)
        du2 = o.fusedg_stage_bwd(g, thin, b, bits, frags, None, dtp, False)
        assert torch.equal(du2, du)


@pytest.mark.parametrize("kw,shape,masked,want_lay", [
    # internal code layout: "rsc" wherever the matrix-core filter-gradient kernel takes the shape -- M > 64 from 2 tiles of
    # 64 x 32 code pixels on (row parts of a tile spread over workgroups), otherwise from 64 tiles on
    (dict(K=4, M=169, P=7, s=2, C=1), (2, 1, 75, 77), False, "rsc"),      # odd size: stride padding in pre_process; 4 tiles
    (dict(K=3, M=64, P=5, s=2, C=1), (1, 1, 40, 72), True, "nchw"),
    (dict(K=3, M=100, P=7, s=1, C=1), (1, 1, 33, 70), False, "rsc"),
    (dict(K=3, M=169, P=7, s=2, C=1), (16, 1, 128, 250), True, "rsc"),
    (dict(K=3, M=72, P=5, s=1, C=1), (12, 1, 96, 100), False, "rsc"),     # unit stride, ragged last strip
    (dict(K=3, M=48, P=7, s=1, C=1), (2, 1, 40, 60), False, "nchw"),      # few tiles, M <= 64: VALU filter gradients
])
def test_strip_sweeps_equal_generic_on_same_activations(kw, shape, masked, want_lay):
    """Forward: strip sweep vs generic sweep; reverse: both sweeps fed the SAME saved activations (no support flip can
    enter), every gradient to split-bf16 accuracy; and the strip sweep is reproducible bit for bit."""
    import cdlnet_video_amd as cva
    from cdlnet_video_amd import loop
    o = cva.ops
    torch.manual_seed(21)
    K, M, P, s = kw["K"], kw["M"], kw["P"], kw["s"]
    net = cva.CDLNet(**kw, t0=5e-3, adaptive=True, init=True)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_ == "t":
                p_.uniform_(2e-3, 2e-2)
            elif n_ != "g":
                p_.add_(0.05 * p_.abs().mean() * torch.randn_like(p_))
    net = net.cuda()
    x = cva.utils.synthetic_clip(shape, seed=5)
    y, sig = cva.awgn(x, (20, 30), torch.Generator().manual_seed(6))
    mask = (torch.rand(x.shape, generator=torch.Generator().manual_seed(7)) < 0.6).float().cuda() if masked else None
    yd = y.cuda() if mask is None else (mask * y.cuda())
    yp, mean, pads, mask_p = o.preprocess(yd, s, mask)
    N = shape[0]
    g = o.Geometry.make(N, 1, M, yp.shape[2:], (P, P), (P // 2, P // 2), s)
    assert o.fusedg_supported(g) and not o.fused_supported(g)
    c = (sig.reshape(-1) / 255.0).cuda()
    tau = o.thresholds(net.t.detach(), c, N)
    A = [m.weight.detach() for m in net.A]
    B = [m.weight.detach() for m in net.B]
    lay = o.fusedg_code_layout(g)                      # "rsc" when the matrix-core filter-gradient kernel takes the shape
    assert lay == want_lay
    xp, z, codes, resid, maps = loop._forward_fusedg(g, yp, mask_p, tau, A, B, True, True, lay)
    xpg, zg, codes_g, resid_g, _ = loop._forward_generic(g, yp, mask_p, tau, A, B, True, True)
    tag = f"strip sweep[{lay}] K{K} M{M} P{P} s{s} {shape}"
    check(f"{tag} xp", xp, xpg, 1e-5)
    check(f"{tag} z_K", z, zg, 5e-5)
    codes_n = [zc if (lay == "nchw" or i == K - 1) else o.fusedg_from_rsc(g, zc) for i, zc in enumerate(codes)]
    assert all(torch.equal(m, o.fusedg_support_map(g, zc)) for m, zc in zip(maps, codes_n))
    xp2, z2, _, _, _ = loop._forward_fusedg(g, yp, mask_p, tau, A, B, False, False, lay)     # ping-pong buffers
    assert torch.equal(xp2, xp) and torch.equal(z2, z)
    if lay != "nchw":                                   # the layout changes no value
        xp3, z3, codes3, _, _ = loop._forward_fusedg(g, yp, mask_p, tau, A, B, True, True, "nchw")
        assert torch.equal(xp3, xp) and torch.equal(z3, z) and all(torch.equal(a, b) for a, b in zip(codes3, codes_n))
    g_xp = torch.randn(xp.shape, generator=torch.Generator().manual_seed(8)).cuda()
    g_z = torch.randn(z.shape, generator=torch.Generator().manual_seed(9)).cuda() * 0.01
    outs = {}
    for name, sweep in (("strip", loop._backward_fusedg), ("generic", loop._backward_generic), ("again", loop._backward_fusedg)):
        dt = torch.zeros(K, 2, M, device="cuda")
        if name == "generic":
            dA, dB = sweep(g, K, yp, mask_p, c, A, B, codes_n, resid, g_xp, g_z, dt, maps=maps)
        else:
            dA, dB = sweep(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=maps, layout=lay)
        outs[name] = (dA, dB, dt)
    for k in range(K):
        check(f"{tag} dA[{k}]", outs["strip"][0][k], outs["generic"][0][k], 5e-5)
        check(f"{tag} dB[{k}]", outs["strip"][1][k], outs["generic"][1][k], 5e-5)
        assert torch.equal(outs["strip"][0][k], outs["again"][0][k]) and torch.equal(outs["strip"][1][k], outs["again"][1][k])
    check(f"{tag} dt", outs["strip"][2], outs["generic"][2], 5e-5)
    assert torch.equal(outs["strip"][2], outs["again"][2])
    # the reverse sweep without the forward's maps rebuilds them from the codes: same result
    dt = torch.zeros(K, 2, M, device="cuda")
    dA, dB = loop._backward_fusedg(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=None, layout=lay)
    assert all(torch.equal(a, b) for a, b in zip(dA, outs["strip"][0])) and torch.equal(dt, outs["strip"][2])


@pytest.mark.parametrize("N,M,P,s,sp", [(2, 169, 7, 2, (64, 128)), (1, 100, 7, 1, (33, 70)), (1, 169, 7, 2, (150, 154))])
def test_strip_rsc_layout_is_bit_identical_to_the_reference_layout(N, M, P, s, sp):
    """CDL_LAY_RSC (row-strip channel-major, what the sweeps keep their internal codes in) against the reference's
    (N,M,Hz,Wz): same values bit for bit, on either side of the forward and of the reverse stage."""
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(17 + M)
    g = make_geom(N, M, P, s, sp)
    r = torch.randn(g.image_shape(), generator=gen).cuda()
    z = (torch.randn(g.code_shape(), generator=gen) * (torch.rand(g.code_shape(), generator=gen) < 0.3)).cuda()
    w = (torch.randn(g.filter_shape(), generator=gen) * 0.1).cuda()
    tau = (torch.rand(N, M, generator=gen) * 0.6 + 0.01).cuda()
    frags, patches = o.fusedg_prep(g, w, w), o.fusedg_patches(g, "cuda")
    assert torch.equal(o.fusedg_from_rsc(g, o.fusedg_to_rsc(g, z)), z)
    ref = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches)
    r_ref = o.fusedg_assemble(g, patches)
    zr = o.fusedg_to_rsc(g, z)
    for li, lo in (("rsc", "rsc"), ("rsc", "nchw"), ("nchw", "rsc")):
        out = o.fusedg_iter(g, r, zr if li == "rsc" else z, tau, frags, -1.0, patches, lay_in=li, lay_out=lo)
        assert torch.equal(out if lo == "nchw" else o.fusedg_from_rsc(g, out), ref), (li, lo)
        assert torch.equal(o.fusedg_assemble(g, patches), r_ref)
    bits = o.fusedg_support_map(g, z)
    dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
    base = torch.randn(g.code_shape(), generator=gen).cuda()
    du_ref = o.fusedg_stage_bwd(g, r, base, bits, frags, patches, dtp, True)
    dtp_ref = dtp.clone()
    for li, lo in (("rsc", "rsc"), ("nchw", "rsc")):
        du = o.fusedg_stage_bwd(g, r, o.fusedg_to_rsc(g, base) if li == "rsc" else base, bits, frags, patches, dtp, True,
                                lay_in=li, lay_out=lo)
        assert torch.equal(o.fusedg_from_rsc(g, du), du_ref) and torch.equal(dtp, dtp_ref), (li, lo)


def test_s2030_net_routes_through_the_strip_kernel_and_is_sample_independent():
    """The shipped architecture at a batch: the fused path is what runs (no silent fall-back to three launches per
    iteration), batched == single-sample bit for bit."""
    import cdlnet_video_amd as cva
    torch.manual_seed(2)
    net = cva.CDLNet(K=6, M=169, P=7, s=2, C=1, t0=5e-3, adaptive=True, init=True).cuda()
    x = cva.utils.synthetic_clip((5, 1, 96, 128), seed=2)
    y, sig = cva.awgn(x, (20, 30), torch.Generator().manual_seed(3))
    g = cva.ops.Geometry.make(5, 1, 169, (96, 128), (7, 7), (3, 3), 2)
    assert cva.ops.fusedg_supported(g) and cva.loop.BACKEND == "auto"
    # the strip layout in both modes: forward-only sweeps always, training sweeps because the matrix-core filter-gradient
    # kernel takes M = 169 from 2 tiles on (10 here); a one-tile geometry keeps the reference layout for training
    assert cva.ops.fusedg_code_layout(g, training=True) == "rsc" and cva.ops.fusedg_code_layout(g, training=False) == "rsc"
    g1 = cva.ops.Geometry.make(1, 1, 169, (64, 128), (7, 7), (3, 3), 2)
    assert cva.ops.fusedg_code_layout(g1, training=True) == "nchw" and cva.ops.fusedg_code_layout(g1, training=False) == "rsc"
    with torch.no_grad():
        xhat, z = net(y.cuda(), sig.cuda())
        for n in (0, 4):
            xn, zn = net(y[n:n + 1].cuda(), sig[n:n + 1].cuda())
            assert torch.equal(xn, xhat[n:n + 1]) and torch.equal(zn, z[n:n + 1]), n
    assert torch.isfinite(xhat).all()
    xt, zt = net(y.cuda(), sig.cuda())                    # the training-mode forward: same values
    assert torch.equal(xt.detach(), xhat) and torch.equal(zt.detach(), z)
    log(f"s2030-arch K6 batch 5x96x128: PSNR noisy {cva.psnr(x, y):.2f} -> {cva.psnr(x, xhat.cpu()):.2f}")


def test_strip_assemble_vector_form_is_bit_identical(hip_env):
    """k_assemble_s<.., 4> (W % 4 == 0: 4 pixels per thread, 16-byte thin accesses) against the scalar form
    (CDL_SCALAR_ASSEMBLE=1): same sums in the same order, stride 1 and 2."""
    import cdlnet_video_amd as cva
    o = cva.ops
    for N, M, P, s, sp in ((3, 169, 7, 2, (72, 136)), (2, 100, 7, 1, (40, 68)), (2, 64, 5, 2, (40, 72))):
        g = make_geom(N, M, P, s, sp)
        gen = torch.Generator(device="cuda").manual_seed(9)
        patches = torch.randn(o.fusedg_patches(g, "cuda").shape, device="cuda", generator=gen)
        yp = torch.randn(g.image_shape(), device="cuda", generator=gen)
        mask = (torch.rand(g.image_shape(), device="cuda", generator=gen) < 0.5).float()
        outs = []
        for scalar in ("0", "1"):
            hip_env("CDL_SCALAR_ASSEMBLE", scalar)
            outs.append((o.fusedg_assemble(g, patches, mask, yp, 1.0), o.fusedg_assemble(g, patches, None, None, -1.0)))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (N, M, P, s, sp)
