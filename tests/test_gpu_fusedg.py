"""The fused iteration for C > 1 / 3-D / P in {3,5,7} / M <= 64 (cdl_fusedg_*) against the shape-generic kernels."""
import pytest
import torch

from gpu_util import check, log

pytestmark = pytest.mark.gpu

# N, C, M, spatial, P (per axis), masked
SHAPES = [
    (1, 1, 48, (8, 16, 64), (5, 5, 5), False),        # cfg3's operator, exactly one tile per depth
    (2, 1, 48, (5, 21, 70), (5, 5, 5), False),        # ragged tiles, depth 5 = Pd
    (1, 3, 64, (30, 70), (7, 7), True),               # cfg4's operator (JDD): C = 3 + mask, 2-D
    (2, 1, 24, (3, 17, 33), (3, 3, 3), False),        # P = 3, M not a multiple of 16
    (1, 1, 8, (40, 72), (5, 5), False),               # small M, 2-D, one group
    (1, 1, 64, (4, 16, 64), (3, 7, 7), True),         # Pd = 3 with 7 x 7 planes
    (1, 1, 16, (2, 20, 36), (5, 5, 5), False),        # fewer frames than depth taps: most planes of a tile are absent
]


@pytest.fixture(params=["tile", "strip"])
def kernel(request, hip_env):
    """Both kernels behind the cdl_fusedg_* entry points for these shapes: the tile kernel k_stage_g (the default) and
    the same arithmetic in the strip decomposition (cdl_stripg.hip, CDL_FUSEDG_STRIP=1)."""
    if request.param == "strip":
        hip_env("CDL_FUSEDG_STRIP", "1")
    return request.param


def make_geom(N, C, M, sp, P):
    import cdlnet_video_amd as cva
    return cva.ops.Geometry.make(N, C, M, sp, P, tuple(p // 2 for p in P), 1)


@pytest.mark.parametrize("N,C,M,sp,P,masked", SHAPES)
def test_fusedg_iteration_vs_generic(N, C, M, sp, P, masked, kernel):
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(sum(sp) + M)
    g = make_geom(N, C, M, sp, P)
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
    tag = f"fusedg[{kernel}] N{N}C{C}M{M} {sp} P{P}"
    for name, zin, sgn in (("iter", z, -1.0), ("first", None, 1.0)):
        z_ref = o.analysis(g, r, wA, sgn, zin, None, tau)
        r_ref = o.synthesis(g, z_ref, wB, 1.0, None, mask, yp)
        patches.fill_(float("nan"))                       # every patch word that is read must be written
        bits = torch.full((N, 4) + tuple(g.dims), -1, dtype=torch.int32, device="cuda")
        z_got = o.fusedg_iter(g, r, zin, tau, frags, sgn, patches, map_out=bits)
        assert torch.equal(bits, o.fusedg_support_map(g, z_got)), f"{tag} {name}: support/sign map"
        r_got = o.fusedg_assemble(g, patches, mask, yp)
        check(f"{tag} {name} z'", z_got, z_ref, 2e-5)
        check(f"{tag} {name} r_next", r_got, o.synthesis(g, z_got, wB, 1.0, None, mask, yp), 2e-5)
        check(f"{tag} {name} r_next(end-to-end)", r_got, r_ref, 8e-5)
        assert float(((z_got != 0) == (z_ref != 0)).float().mean()) > 0.9999
        z2 = o.fusedg_iter(g, r, zin, tau, frags, sgn, patches)          # deterministic, with or without the map
        assert torch.equal(z2, z_got)


@pytest.mark.parametrize("N,C,M,sp,P,masked", SHAPES[:4])
def test_fusedg_backward_stage_vs_generic(N, C, M, sp, P, masked, kernel):
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(7 * sum(sp) + M)
    g = make_geom(N, C, M, sp, P)
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
    tiles = o._fusedg_sizes(g)[2]
    dtp = torch.empty(tiles, M, device="cuda")
    bits = o.fusedg_support_map(g, gate)
    tag = f"fusedg-bwd[{kernel}] N{N}C{C}M{M} {sp} P{P}"
    for name, b in (("with-base", base), ("no-base", None)):
        gk = o.analysis(g, thin, w1, 1.0, b, None, None)
        du_ref = gk * (gate != 0)
        dt_ref = torch.zeros(2, M, device="cuda")
        o.tau_grad(g, gk, gate, c, dt_ref)
        q_ref = o.synthesis(g, gk, w2, -1.0, gate, mask, None)
        patches.fill_(float("nan"))
        du = o.fusedg_stage_bwd(g, thin, b, bits, frags, patches, dtp, True)
        q = o.fusedg_assemble(g, patches, mask, None, -1.0)
        dt = torch.zeros(2, M, device="cuda")
        o.fusedg_dtau_reduce(g, dtp, c, dt)
        check(f"{tag} {name} du", du, du_ref, 2e-5)
        differ = (du == 0) != (du_ref == 0)             # same support, bar an entry the two arithmetics cancel differently
        assert int(differ.sum()) <= 2 and float((du - du_ref)[differ].abs().max() if differ.any() else 0.0) < 1e-4
        check(f"{tag} {name} q", q, q_ref, 8e-5)
        check(f"{tag} {name} dt", dt, dt_ref, 8e-5)
        du2 = o.fusedg_stage_bwd(g, thin, b, bits, frags, None, dtp, False)
        assert torch.equal(du2, du)


@pytest.mark.parametrize("kind,kw,shape,masked", [
    ("3d", dict(K=4, M=48, P=[5, 5, 5], s=1, C=1), (2, 1, 6, 20, 40), False),
    ("2d", dict(K=5, M=64, P=7, s=1, C=3), (1, 3, 33, 70), True),
    ("2d", dict(K=3, M=16, P=3, s=1, C=1), (2, 1, 24, 24), False),
])
def test_fusedg_sweeps_equal_generic_on_same_activations(kind, kw, shape, masked, kernel):
    """Forward: fused vs generic sweep (1e-5 on xhat-like outputs); reverse: both sweeps fed the SAME saved
    activations (no support flip can enter), every gradient to split-bf16 accuracy; and the fused sweep is
    reproducible bit for bit."""
    import cdlnet_video_amd as cva
    from cdlnet_video_amd import loop
    o = cva.ops
    torch.manual_seed(21)
    K, M, P = kw["K"], kw["M"], kw["P"]
    if kind == "3d":
        net = cva.CDLNetVideo(**kw, t0=5e-3, adaptive=True, depth=shape[2], init=True)
    else:
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
    mask = cva.gen_bayer_mask(x).cuda() if masked else None
    yd = y.cuda() if mask is None else (mask * y.cuda())
    yp, mean, pads, mask_p = o.preprocess(yd, 1, mask)
    N = shape[0]
    Pt = tuple(P) if isinstance(P, (list, tuple)) else (P, P)
    g = o.Geometry.make(N, shape[1], M, yp.shape[2:], Pt, tuple(p // 2 for p in Pt), 1)
    assert o.fusedg_supported(g)
    c = (sig.reshape(-1) / 255.0).cuda()
    tau = o.thresholds(net.t.detach(), c, N)
    A = [m.weight.detach() for m in net.A]
    B = [m.weight.detach() for m in net.B]
    xp, z, codes, resid, maps = loop._forward_fusedg(g, yp, mask_p, tau, A, B, True, True)
    xpg, zg, codes_g, resid_g, _ = loop._forward_generic(g, yp, mask_p, tau, A, B, True, True)
    tag = f"fusedg sweep[{kernel}] {kind} K{K} M{M} P{P} {shape}"
    check(f"{tag} xp", xp, xpg, 1e-5)
    check(f"{tag} z_K", z, zg, 5e-5)
    assert all(torch.equal(m, o.fusedg_support_map(g, zc)) for m, zc in zip(maps, codes))
    xp2, z2, _, _, _ = loop._forward_fusedg(g, yp, mask_p, tau, A, B, False, False)          # ping-pong buffers
    assert torch.equal(xp2, xp) and torch.equal(z2, z)
    g_xp = torch.randn(xp.shape, generator=torch.Generator().manual_seed(8)).cuda()
    g_z = torch.randn(z.shape, generator=torch.Generator().manual_seed(9)).cuda() * 0.01
    outs = {}
    for name, sweep in (("fusedg", loop._backward_fusedg), ("generic", loop._backward_generic), ("again", loop._backward_fusedg)):
        dt = torch.zeros(K, 2, M, device="cuda")
        dA, dB = sweep(g, K, yp, mask_p, c, A, B, codes, resid, g_xp, g_z, dt, maps=maps)
        outs[name] = (dA, dB, dt)
    for k in range(K):
        check(f"{tag} dA[{k}]", outs["fusedg"][0][k], outs["generic"][0][k], 5e-5)
        check(f"{tag} dB[{k}]", outs["fusedg"][1][k], outs["generic"][1][k], 5e-5)
        assert torch.equal(outs["fusedg"][0][k], outs["again"][0][k]) and torch.equal(outs["fusedg"][1][k], outs["again"][1][k])
    check(f"{tag} dt", outs["fusedg"][2], outs["generic"][2], 5e-5)
    assert torch.equal(outs["fusedg"][2], outs["again"][2])


def test_fusedg_assemble_vector_form_is_bit_identical(hip_env):
    """k_assemble_g4 (W % 4 == 0: 4 pixels per thread, 16-byte thin accesses) against the scalar form
    (CDL_SCALAR_ASSEMBLE=1): same sums in the same order."""
    import cdlnet_video_amd as cva
    o = cva.ops
    for N, C, M, sp, P in ((2, 1, 48, (6, 40, 72), (5, 5, 5)), (1, 3, 64, (36, 132), (7, 7))):
        g = make_geom(N, C, M, sp, P)
        gen = torch.Generator(device="cuda").manual_seed(9)
        patches = torch.randn(o.fusedg_patches(g, "cuda").shape, device="cuda", generator=gen)
        yp = torch.randn(g.image_shape(), device="cuda", generator=gen)
        mask = (torch.rand(g.image_shape(), device="cuda", generator=gen) < 0.5).float()
        outs = []
        for dbg in ("0", "1"):
            hip_env("CDL_SCALAR_ASSEMBLE", dbg)
            outs.append((o.fusedg_assemble(g, patches, mask, yp, 1.0), o.fusedg_assemble(g, patches, None, None, -1.0)))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_fusedg_timeline_hook_records_and_does_not_disturb(hip_env):
    """cdl_fusedg_set_timeline: with a buffer registered a stage launch stamps increasing s_memtime values for every
    wave of workgroup 0 and computes the same bits; unregistered, nothing is written."""
    import cdlnet_video_amd as cva
    o = cva.ops
    g = make_geom(1, 1, 48, (4, 32, 64), (5, 5, 5))
    gen = torch.Generator(device="cuda").manual_seed(3)
    r = torch.randn(g.image_shape(), device="cuda", generator=gen)
    z = torch.randn(g.code_shape(), device="cuda", generator=gen)
    w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
    tau = torch.full((1, 48), 0.3, device="cuda")
    frags = o.fusedg_prep(g, w, w)
    patches = o.fusedg_patches(g, "cuda")
    ref = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches).clone()
    pref = patches.clone()
    tl = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
    lib = cva._lib.lib()
    try:
        lib.cdl_fusedg_set_timeline(tl.data_ptr())
        got = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches).clone()
        torch.cuda.synchronize()
    finally:
        lib.cdl_fusedg_set_timeline(None)
    assert torch.equal(got, ref) and torch.equal(patches, pref)
    t = tl.view(8, 256).cpu()
    n = int((t[0] > 0).sum())
    assert n >= 21                                               # kernel start + 20 stamps of the first tile
    assert bool((t[:, 1:n] >= t[:, :n - 1]).all())
    tl.zero_()
    o.fusedg_iter(g, r, z, tau, frags, -1.0, patches)
    torch.cuda.synchronize()
    assert int(tl.abs().sum()) == 0


def test_one_instantiation_serves_a_larger_lds_carving_after_a_smaller_one(kernel):
    """k_stage_g<P,G,MT,MODE> takes its dynamic-LDS size from M (KQ = ceil(M/16)); M = 48 needs
    ~138 KB, M = 64 ~148 KB with the same template arguments.  The launcher must raise the kernel's limit when a
    LARGER request follows a smaller one in the same process (cdl_ensure_dynamic_lds keeps the maximum)."""
    import cdlnet_video_amd as cva
    o = cva.ops
    gen = torch.Generator().manual_seed(11)
    for M in (40, 48, 56, 64):                       # ascending: every step asks for more LDS than the last
        g = make_geom(1, 1, M, (5, 16, 64), (5, 5, 5))
        assert o.fusedg_supported(g)
        r = torch.randn(g.image_shape(), generator=gen).cuda()
        z = (torch.randn(g.code_shape(), generator=gen) * 0.3).cuda()
        w = (torch.randn(g.filter_shape(), generator=gen) * 0.1).cuda()
        tau = torch.full((1, M), 0.05).cuda()
        got = o.fusedg_iter(g, r, z, tau, o.fusedg_prep(g, w, w), -1.0, o.fusedg_patches(g, "cuda"))
        check(f"fusedg LDS growth M{M}", got, o.analysis(g, r, w, -1.0, z, None, tau), 2e-5)
