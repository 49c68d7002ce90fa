"""Operator-level parity: every C-ABI entry point against the CPU oracle on seeded inputs."""
import pytest
import torch

from gpu_util import check
from oracle import cdl_oracle as O

pytestmark = pytest.mark.gpu


def ops():
    import cdlnet_video_amd as cva
    return cva.ops


# (N, C, M, spatial, P, stride)
SHAPES = [
    (2, 1, 8, (20, 24), (5, 5), 1),
    (1, 1, 64, (40, 36), (7, 7), 1),
    (2, 3, 9, (18, 22), (7, 7), 1),
    (2, 1, 6, (24, 20), (7, 7), 2),
    (1, 3, 5, (16, 12), (3, 5), 2),
    (1, 1, 6, (8, 16, 16), (5, 5, 5), 1),
    (2, 1, 5, (8, 12, 20), (9, 9, 5), 2),
    (1, 2, 4, (6, 9, 12), (3, 3, 3), 3),
    (1, 1, 169, (16, 18), (7, 7), 2),
    # enough 64 x 32 code tiles x (C * Pd) for the LDS-tiled filter gradient k_wgrad_l (>= 128 workgroups),
    # with ragged tiles, C = 3, stride 2 and a 3-D filter
    (24, 3, 5, (40, 70), (7, 7), 1),
    (48, 1, 6, (132, 72), (7, 7), 2),
    (3, 1, 5, (9, 40, 72), (5, 5, 5), 1),
    (70, 2, 4, (34, 66), (9, 9), 1),
    # the argscsr.json filter bank: 9 x 9, stride 2, 169 code channels -- weight fragments stream through LDS in
    # chunks of k-steps in the matrix-core synthesis, channel groups in the analysis
    (12, 1, 169, (68, 132), (9, 9), 2),
    # the shipped s2030 filter bank with enough tiles (80) for the matrix-core filter gradient: 6 channel tiles = 3 wave
    # groups x 2 pixel parts, two waves idle in the tree reduction
    (20, 1, 169, (68, 132), (7, 7), 2),
    # many channels on both sides, unit stride: the dense matrix-core tier (cdl_dense_mfma.hip; the ResidualBlock's
    # Conv3d(M, M) and its data gradient), ragged channel chunks (24 = 16 + 8), M over two workgroup groups, 2-D
    (2, 16, 32, (3, 17, 33), (3, 3, 3), 1),
    (1, 24, 40, (20, 45), (3, 3), 1),
    (1, 20, 70, (2, 9, 34), (1, 5, 5), 1),
    # the args3dmri.json filter (9, 9, 5) on the matrix cores: 405 taps over 9 depth planes -- one channel tile per analysis
    # workgroup with > 96 KB of LDS, rectangular 9 x 5 planes in all three kernels, the filter gradient's (c, kd) group passes
    # spread over workgroups (24 / 16 tiles); stride 2 as shipped, and stride 1
    (6, 1, 40, (8, 32, 72), (9, 9, 5), 2),
    (4, 1, 33, (4, 24, 40), (9, 9, 5), 1),
]


def make(N, C, M, sp, P, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C) + sp, generator=g)
    zsp = tuple(d // s for d in sp)
    z = torch.randn((N, M) + zsp, generator=g)
    z = z * (torch.rand(z.shape, generator=g) < 0.3)          # sparse code with exact zeros
    w = torch.randn((M, C) + P, generator=g) * 0.2
    return x, z, w


@pytest.mark.parametrize("path,tol", [("mfma", 2e-5), ("valu", 2e-6)])
@pytest.mark.parametrize("N,C,M,sp,P,s", SHAPES)
def test_analysis_variants(N, C, M, sp, P, s, path, tol, hip_env):
    """Both analysis paths: the matrix-core kernel (default where it exists: >= 96 workgroups of 4 tiles; smaller
    launches fall through) and the fp32 VALU kernels (CDL_MFMA_ANALYSIS=0)."""
    hip_env("CDL_MFMA_ANALYSIS", "1" if path == "mfma" else "0")
    hip_env("CDL_MFMA_DENSE", "1" if path == "mfma" else "0")
    o = ops()
    x, z, w = make(N, C, M, sp, P, s)
    pad = tuple(p // 2 for p in P)
    geom = o.Geometry.make(N, C, M, sp, P, pad, s)
    ref_conv = O.analysis(x, w, s, pad)
    # negative thresholds make ST jump by 2|t| at u = 0: an element whose u changes sign within the 5e-6 of the
    # split-bf16 products would fail any tolerance, so they are exercised on the fp32 path (and by fixture f6)
    tau = torch.rand(N, M) * 0.5 + (0.01 if path == "mfma" else -0.1)
    tb = tau.reshape((N, M) + (1,) * len(sp))
    xd, zd, wd = x.cuda(), z.cuda(), w.cuda()
    tag = f"analysis[{path}] N{N}C{C}M{M}{sp}P{P}s{s}"
    check(tag + " plain", o.analysis(geom, xd, wd), ref_conv, tol)
    check(tag + " first", o.analysis(geom, xd, wd, 1.0, None, None, tau.cuda()),
          O.soft_threshold(ref_conv, tb), 2 * tol)
    check(tag + " iter", o.analysis(geom, xd, wd, -1.0, zd, None, tau.cuda()),
          O.soft_threshold(z - ref_conv, tb), 2 * tol)
    gup = torch.randn(z.shape)
    check(tag + " bwd", o.analysis(geom, xd, wd, 1.0, gup.cuda(), zd, None),
          gup * (z != 0) + ref_conv, tol)


@pytest.mark.parametrize("path,tol", [("mfma", 2e-5), ("valu", 2e-6)])
@pytest.mark.parametrize("N,C,M,sp,P,s", SHAPES)
def test_synthesis_variants(N, C, M, sp, P, s, path, tol, hip_env):
    """Both synthesis paths: the matrix-core kernels (split-bf16 x3; default wherever they exist, the other
    shapes fall through to the VALU kernels) and the fp32 VALU kernels (CDL_MFMA_SYNTHESIS=0; the library snapshots its switches, hip_env reloads them)."""
    hip_env("CDL_MFMA_SYNTHESIS", "1" if path == "mfma" else "0")
    hip_env("CDL_MFMA_DENSE", "1" if path == "mfma" else "0")
    o = ops()
    x, z, w = make(N, C, M, sp, P, s, seed=1)
    pad = tuple(p // 2 for p in P)
    geom = o.Geometry.make(N, C, M, sp, P, pad, s)
    ref = O.synthesis(z, w, s, pad)
    assert ref.shape == x.shape
    mask = (torch.rand(x.shape) < 0.4).float()
    zd, wd = z.cuda(), w.cuda()
    tag = f"synthesis[{path}] N{N}C{C}M{M}{sp}P{P}s{s}"
    check(tag + " plain", o.synthesis(geom, zd, wd), ref, tol)
    check(tag + " resid", o.synthesis(geom, zd, wd, 1.0, None, mask.cuda(), x.cuda()), mask * ref - x, tol)
    gup = torch.randn(z.shape)
    ref_b = -mask * O.synthesis(gup * (z != 0), w, s, pad)
    check(tag + " bwd", o.synthesis(geom, gup.cuda(), wd, -1.0, zd, mask.cuda(), None), ref_b, tol)
    # deterministic: patches are gathered and added in a fixed order
    assert torch.equal(o.synthesis(geom, zd, wd), o.synthesis(geom, zd, wd))


@pytest.mark.parametrize("path", ["mfma", "valu"])
@pytest.mark.parametrize("N,C,M,sp,P,s", SHAPES)
def test_filter_and_threshold_grads(N, C, M, sp, P, s, path, hip_env):
    """Filter gradients through the matrix-core kernel (default where it exists: >= 64 tiles of 64 x 32 code
    pixels; smaller launches fall through) and through the fp32 VALU kernels (CDL_MFMA_WGRAD=0)."""
    hip_env("CDL_MFMA_WGRAD", "1" if path == "mfma" else "0")
    hip_env("CDL_MFMA_DENSE", "1" if path == "mfma" else "0")
    o = ops()
    x, z, w = make(N, C, M, sp, P, s, seed=2)
    pad = tuple(p // 2 for p in P)
    geom = o.Geometry.make(N, C, M, sp, P, pad, s)
    # d/dw of <analysis(x; w), u> = sum u (x) x: autograd on the oracle is the reference
    wv = w.clone().requires_grad_(True)
    u = torch.randn(z.shape)
    (O.analysis(x, wv, s, pad) * (u * (z != 0))).sum().backward()
    tag = f"wgrad[{path}] N{N}C{C}M{M}{sp}P{P}s{s}"
    wtol = 2e-5 if path == "mfma" else 1e-5           # split-bf16 x3 operands (as the fused kernels) vs fp32 FMAs
    check(tag + " gated", o.wgrad(geom, u.cuda(), x.cuda(), 1.0, gate=z.cuda()), wv.grad, wtol)
    g_ana = wv.grad.clone()
    wv.grad = None
    (O.synthesis(z, wv, s, pad) * x).sum().backward()
    check(tag + " synth", o.wgrad(geom, z.cuda(), x.cuda(), -2.0), -2.0 * wv.grad, wtol)
    # the dA_k / dB_k pair of a reverse iteration as one launch (cdl_wgrad_pair; gradient gated upstream)
    d0, d1 = o.wgrad_pair(geom, (u * (z != 0)).cuda(), x.cuda(), 1.0, z.cuda(), x.cuda(), -2.0)
    check(tag + " pair[0]", d0, g_ana, wtol)
    check(tag + " pair[1]", d1, -2.0 * wv.grad, wtol)
    # threshold gradient
    c = torch.rand(N)
    dt = torch.zeros(2, M, device="cuda")
    o.tau_grad(geom, u.cuda(), z.cuda(), c.cuda(), dt)
    s_nm = -(torch.sign(z) * u).sum(dim=tuple(range(2, z.dim())))
    check(tag + " dt0", dt[0], s_nm.sum(0), 1e-5)
    check(tag + " dt1", dt[1], (c[:, None] * s_nm).sum(0), 1e-5)


@pytest.mark.parametrize("shape,s,masked", [((2, 1, 33, 31), 2, False), ((3, 3, 21, 19), 2, True),
                                            ((2, 1, 32, 32), 1, False), ((1, 1, 7, 13, 11), 2, False),
                                            ((2, 3, 17, 20), 4, True), ((1, 2, 5, 6, 7), 4, True)])
def test_pre_and_post_process(shape, s, masked):
    o = ops()
    g = torch.Generator().manual_seed(3)
    y = torch.rand(shape, generator=g)
    mask = (torch.rand(shape, generator=g) < 0.5).float() if masked else None
    if masked:
        y = y * mask
    yp, mean, pads, mask_p = O.preprocess(y, s, mask)
    yp_d, mean_d, pads_d, mask_d = o.preprocess(y.cuda(), s, None if mask is None else mask.cuda())
    assert tuple(pads_d) == tuple(pads)
    tag = f"preprocess {shape} s{s} mask{masked}"
    check(tag + " yp", yp_d, yp, 2e-6)
    check(tag + " mean", mean_d, mean.reshape(-1), 2e-6)
    if masked:
        assert torch.equal(mask_d.cpu(), mask_p)
    xp = torch.randn(yp.shape, generator=g)
    check(tag + " post", o.postprocess(xp.cuda(), mean_d, pads), O.postprocess(xp, mean, pads), 2e-6)
    gx = torch.randn(shape, generator=g)
    full = torch.zeros(yp.shape)
    idx = [slice(None), slice(None)]
    nsp = len(shape) - 2
    for d in range(nsp):
        lo, hi = pads[2 * (nsp - 1 - d)], pads[2 * (nsp - 1 - d) + 1]
        idx.append(slice(lo, yp.shape[2 + d] - hi))
    full[tuple(idx)] = gx
    assert torch.equal(o.postprocess_bwd(gx.cuda(), pads).cpu(), full)


def test_thresholds_and_shrink():
    o = ops()
    g = torch.Generator().manual_seed(4)
    t = torch.rand(5, 2, 7, 1, 1, generator=g) * 0.1
    c = torch.rand(3, generator=g)
    tau = o.thresholds(t.cuda(), c.cuda(), 3).cpu()
    ref = t[:, 0].reshape(5, 1, 7) + c.reshape(1, 3, 1) * t[:, 1].reshape(5, 1, 7)
    assert torch.equal(tau, ref)
    assert torch.equal(o.thresholds(t.cuda(), None, 3).cpu(), t[:, 0].reshape(5, 1, 7).expand(5, 3, 7))
    x = torch.randn(3, 7, 9, 11, generator=g) * 0.1
    x[0, 0, 0, :3] = torch.tensor([0.0, -0.0, 1e-30])
    tt = torch.rand(3, 7, generator=g) * 0.2 - 0.05
    got = o.shrink(x.cuda(), tt.cuda()).cpu()
    assert torch.equal(got, O.soft_threshold(x, tt.reshape(3, 7, 1, 1)))
    import cdlnet_video_amd as cva
    got = cva.ST(x.cuda(), tt.reshape(3, 7, 1, 1).cuda()).cpu()
    assert torch.equal(got, O.soft_threshold(x, tt.reshape(3, 7, 1, 1)))


def test_golden_shrink_table(golden):
    o = ops()
    g = golden("f6_negative_t")
    x = g["st_in"].reshape(1, 1, -1)
    for row, t in zip(g["st_out"], g["st_t"].tolist()):
        got = o.shrink(x.cuda(), torch.tensor([[t]], dtype=torch.float32).cuda()).cpu().reshape(-1)
        assert torch.equal(got, row)


def test_project(golden):
    o = ops()
    g = golden("f9_helpers")
    w = g["W"].clone().cuda()
    o.project_filters_(w)
    check("project 2d vs reference uball_project", w, g["W_proj"], 1e-6)
    w3 = g["W3"].clone().cuda()
    o.project_filters_(w3)
    check("project 3d vs oracle (parity unpinned: reference call raises)", w3,
          O.unit_ball(g["W3"], (2, 3, 4)), 1e-6)
    zero = torch.zeros(2, 1, 3, 3, device="cuda")
    o.project_filters_(zero)
    assert torch.equal(zero.cpu(), torch.zeros(2, 1, 3, 3))


@pytest.mark.parametrize("order,M,C,P,tr", [(1, 6, 1, 5, False), (2, 4, 1, 7, True), (3, 5, 3, 9, False)])
def test_gabor_bank_and_adjoint(order, M, C, P, tr):
    o = ops()
    g = torch.Generator().manual_seed(5)
    alpha = torch.randn(order, M, C, 1, 1, generator=g)
    a = torch.randn(order, M, C, 2, generator=g) * 0.4
    w0 = torch.randn(order, M, C, 2, generator=g)
    psi = torch.randn(order, M, C, generator=g)
    leaves = [v.clone().requires_grad_(True) for v in (alpha, a, w0, psi)]
    ref = O.gabor_bank(*leaves, P, tr)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    got = o.gabor_filters(alpha.cuda(), a.cuda(), w0.cuda(), psi.cuda(), P, tr)
    tag = f"gabor o{order}M{M}C{C}P{P}T{tr}"
    check(tag + " fwd", got, ref, 1e-5)
    grads = o.gabor_filters_bwd(alpha.cuda(), a.cuda(), w0.cuda(), psi.cuda(), up.cuda(), P, tr)
    for name, gg, leaf in zip(("dalpha", "da", "dw0", "dpsi"), grads, leaves):
        check(f"{tag} {name}", gg, leaf.grad, 2e-5)


def test_golden_gabor_filters(golden):
    o = ops()
    g = golden("f5_gabor_shared")
    for k in range(3):
        for bank, tr in (("A", True), ("B", False)):
            got = o.gabor_filters(*(g["sd"][f"{bank}.{k}.{f}"].cuda() for f in ("alpha", "a", "w0", "psi")), 7, tr)
            check(f"golden gabor filter {bank}.{k}", got, g["filt"][f"{bank}.{k}"], 1e-5)


def test_batched_gabor_banks_equal_per_bank_calls():
    """cdl_gabor_filter_banks(+_bwd): every bank of a net from one launch -- bit-identical to the per-bank entry
    points, with a missing upstream gradient giving zeros and ALIASED parameters (GDLNet `shared`) summed by autograd."""
    import cdlnet_video_amd as cva
    from cdlnet_video_amd.gabor import ConvAdjoint2dGabor, filter_banks
    o = ops()
    torch.manual_seed(3)
    mods = [ConvAdjoint2dGabor(6, 1, 7, stride=1, order=2).cuda() for _ in range(5)]
    mods[3].psi, mods[3].a = mods[0].psi, mods[0].a                     # parameter sharing by aliasing
    tr = [True, True, False, True, False]
    banks = filter_banks(mods, tr)
    for m, t, w in zip(mods, tr, banks):
        assert torch.equal(w, o.gabor_filters(m.alpha, m.a, m.w0, m.psi, 7, t))
    ups = [torch.randn_like(w) for w in banks]
    loss = sum((w * u).sum() for k, (w, u) in enumerate(zip(banks, ups)) if k != 2)     # bank 2 gets no gradient
    loss.backward()
    per = [o.gabor_filters_bwd(m.alpha, m.a, m.w0, m.psi, u, 7, t) for m, t, u in zip(mods, tr, ups)]
    for k, m in enumerate(mods):
        if k == 2:
            assert m.alpha.grad is None or float(m.alpha.grad.abs().max()) == 0.0
            continue
        check(f"batched gabor dalpha[{k}]", m.alpha.grad, per[k][0], 1e-6)
        check(f"batched gabor dw0[{k}]", m.w0.grad, per[k][2], 1e-6)
    check("batched gabor dpsi (aliased 0+3)", mods[0].psi.grad, per[0][3] + per[3][3], 1e-6)
    check("batched gabor da (aliased 0+3)", mods[0].a.grad, per[0][1] + per[3][1], 1e-6)
    check("batched gabor dpsi[1]", mods[1].psi.grad, per[1][3], 1e-6)


def test_bad_arguments_fail_loudly():
    import cdlnet_video_amd as cva
    o = cva.ops
    with pytest.raises(RuntimeError):
        o.preprocess(torch.rand(1, 1, 8, 8), 1)                     # CPU tensor
    with pytest.raises(ValueError):
        o.Geometry.make(1, 1, 4, (8, 8), (4, 4), (2, 2), 1)         # even filter
    with pytest.raises(ValueError):
        o.Geometry.make(1, 1, 4, (9, 8), (3, 3), (1, 1), 2)         # extent not a stride multiple
    geom = o.Geometry.make(1, 1, 4, (8, 8), (3, 3), (1, 1), 1)
    z = torch.zeros(geom.code_shape(), device="cuda")
    x = torch.zeros(geom.image_shape(), device="cuda")
    w = torch.zeros(geom.filter_shape(), device="cuda")
    with pytest.raises(cva.HipKernelError):
        o.analysis(geom, x, w, 1.0, z, None, None, out=z)           # out aliases zin


def test_synthesis_assemble_vector_form_is_bit_identical(hip_env):
    """k_synth_assemble4 (W % 4 == 0: 4 pixels per thread, 16-byte thin accesses) against the scalar form
    (CDL_SCALAR_ASSEMBLE=1): same terms in the same order, stride 1 and 2, 2-D and 3-D."""
    o = ops()
    for N, C, M, sp, P, s in ((24, 3, 5, (40, 72), (7, 7), 1), (48, 1, 6, (132, 72), (7, 7), 2), (3, 1, 5, (9, 40, 72), (5, 5, 5), 1)):
        x, z, w = make(N, C, M, sp, P, s, seed=5)
        geom = o.Geometry.make(N, C, M, sp, P, tuple(p // 2 for p in P), s)
        mask = (torch.rand(x.shape) < 0.5).float().cuda()
        outs = []
        for dbg in ("0", "1"):
            hip_env("CDL_SCALAR_ASSEMBLE", dbg)
            outs.append(o.synthesis(geom, z.cuda(), w.cuda(), -1.5, None, mask, x.cuda()))
        assert torch.equal(outs[0], outs[1]), (N, C, M, sp, P, s)


@pytest.mark.parametrize("path", ["mfma", "valu"])
@pytest.mark.parametrize("N,C,M,sp,P,s", [SHAPES[1], SHAPES[3], SHAPES[5], SHAPES[10], SHAPES[11], SHAPES[12], SHAPES[14], SHAPES[18]])
def test_reverse_analysis_step(N, C, M, sp, P, s, path, hip_env):
    """cdl_analysis_rev_ws: out = [zsup != 0] (zin + alpha A x) with the threshold gradients of `out` -- fused into the
    matrix-core analysis epilogue where that kernel exists, composed (analysis, then gate + threshold pass) elsewhere --
    against the oracle."""
    hip_env("CDL_MFMA_ANALYSIS", "1" if path == "mfma" else "0")
    hip_env("CDL_MFMA_DENSE", "0")
    o = ops()
    x, z, w = make(N, C, M, sp, P, s, seed=4)
    pad = tuple(p // 2 for p in P)
    geom = o.Geometry.make(N, C, M, sp, P, pad, s)
    gup = torch.randn(z.shape)
    c = torch.rand(N)
    want = (gup + 0.5 * O.analysis(x, w, s, pad)) * (z != 0)
    dt = torch.zeros(2, M, device="cuda")
    got = o.analysis_rev(geom, x.cuda(), w.cuda(), 0.5, gup.cuda(), z.cuda(), c.cuda(), dt)
    tol = 2e-5 if path == "mfma" else 2e-6
    tag = f"analysis_rev[{path}] N{N}C{C}M{M}{sp}P{P}s{s}"
    check(tag + " out", got, want, tol)
    assert bool((got[(z == 0).cuda()] == 0).all())
    s_nm = -(torch.sign(z) * want).sum(dim=tuple(range(2, z.dim())))
    check(tag + " dt0", dt[0], s_nm.sum(0), 2e-5)
    check(tag + " dt1", dt[1], (c[:, None] * s_nm).sum(0), 2e-5)
    dt2 = torch.zeros(2, M, device="cuda")
    got2 = o.analysis_rev(geom, x.cuda(), w.cuda(), 0.5, gup.cuda(), z.cuda(), c.cuda(), dt2)
    assert torch.equal(got, got2) and torch.equal(dt, dt2)              # deterministic
