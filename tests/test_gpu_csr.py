"""CSR temporal variants on the GPU: the HIP proximal maps and the
product modules CDLNet_CSR / CDLNet_CSRf2 against the reference's own outputs (tests/golden/c*.npz,
generated from the unmodified reference by tools/make_golden_csr.py) and the CPU oracle."""
import pytest
import torch

from gpu_util import check, load_golden, log
from oracle import cdl_oracle as O

pytestmark = pytest.mark.gpu

XTOL = 1e-5
GTOL = 2e-4


def build(g, cls_name):
    import cdlnet_video_amd as cva
    K, M, P, s, C = g["hyper"]
    net = getattr(cva, cls_name)(K=K, M=M, P=P, s=s, C=C, t0=0.0, adaptive=True, init=False)
    net.load_state_dict(g["sd"])
    return net.cuda()


def dev(v):
    return v.cuda() if torch.is_tensor(v) else v


def check_param_grads(name, net, g):
    seen = 0
    for pname, p in net.named_parameters():
        ref = g["grad"].get(pname)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        assert p.grad is not None, pname
        check(f"{name} grad {pname}", p.grad, ref, GTOL)
        seen += 1
    assert seen == len(g["grad"])


# ---------------------------------------------------------------------------------- pointwise maps
def test_prox_maps_bit_exact_on_reference_grid():
    """Exact zeros, ties and negative thresholds: the maps are discontinuous there, so equality of
    every output bit with the reference is the only meaningful check."""
    import cdlnet_video_amd as cva
    g = load_golden("c0_prox_pointwise")
    shape = (1, 1, g["u"].numel(), 1)
    u, zp, za = (g[k].reshape(shape).cuda() for k in ("u", "zp", "za"))
    for i, (lam, g1, g2) in enumerate(g["cases"].reshape(-1, 3).tolist()):
        p1 = cva.prox_CSR(u, zp, lam, g1)
        p2 = cva.prox_CSR_f2(u, zp, za, lam, g1, g2)
        assert torch.equal(p1.cpu().reshape(-1), g["prox_csr"][i]), f"prox_CSR case {i}"
        assert torch.equal(p2.cpu().reshape(-1), g["prox_csr_f2"][i]), f"prox_CSR_f2 case {i}"


@pytest.mark.parametrize("both", [False, True])
@pytest.mark.parametrize("shape", [(2, 5, 9, 13), (1, 3, 4, 20, 24), (3, 64, 40, 40)])
def test_prox_forward_and_reverse_vs_oracle(shape, both):
    """Random codes with per-(sample, channel) thresholds: forward bit-exact, reverse against autograd
    of the oracle expression (gu, neighbour gradients accumulated on top of a seed, threshold sums)."""
    from cdlnet_video_amd import ops
    gen = torch.Generator().manual_seed(sum(shape) + both)
    N, M = shape[:2]
    sp = shape[2:]
    u = 0.05 * torch.randn(shape, generator=gen)
    zp = 0.05 * torch.randn(shape, generator=gen) * (torch.rand(shape, generator=gen) > 0.4)
    za = 0.05 * torch.randn(shape, generator=gen) * (torch.rand(shape, generator=gen) > 0.4)
    gz = torch.randn(shape, generator=gen)
    c = torch.rand(N, generator=gen)
    t, g1, g2 = (torch.rand(1, 2, M, generator=gen) * s for s in (0.02, 1.2, 1.2))
    bshape = (N, M) + (1,) * len(sp)
    cb = c.reshape((N,) + (1,) * (len(sp) + 1))
    leaves = [v.clone().requires_grad_(True) for v in (u, zp, za, t, g1, g2)]
    lu, lzp, lza, lt, lg1, lg2 = leaves
    th = lambda p: (p[0, 0].reshape((1, M) + (1,) * len(sp)) + cb * p[0, 1].reshape((1, M) + (1,) * len(sp)))
    ref = (O.prox_csr_f2(lu, lzp, lza, th(lt), th(lg1), th(lg2)) if both else O.prox_csr(lu, lzp, th(lt), th(lg1)))
    (ref * gz).sum().backward()

    geom = ops.Geometry.make(N, 1, M, sp, (1,) * len(sp), (0,) * len(sp), 1)
    cd = c.cuda()
    lam, gam1, gam2 = (ops.thresholds(p.cuda().reshape(1, 2, M, 1, 1), cd, N)[0] for p in (t, g1, g2))
    out = ops.prox_csr(geom, u.cuda(), zp.cuda(), lam, gam1, za.cuda() if both else None, gam2 if both else None)
    assert torch.equal(out.cpu(), ref.detach()), "forward map differs from the oracle"

    seed = torch.randn(shape, generator=gen)
    gzp, gza = seed.cuda().clone(), seed.cuda().clone()
    dl, d1, d2 = (torch.zeros(2, M, device="cuda") for _ in range(3))
    gu = ops.prox_csr_bwd(geom, gz.cuda(), u.cuda(), zp.cuda(), lam, gam1, cd, dl, d1,
                          za.cuda() if both else None, gam2 if both else None, d2 if both else None,
                          gzp, gza if both else None)
    tag = f"prox{'_f2' if both else ''} {shape}"
    check(f"{tag} gu", gu, lu.grad, 1e-6)
    check(f"{tag} gz_prev", gzp.cpu() - seed, lzp.grad, 1e-5)
    check(f"{tag} dlam", dl, lt.grad[0], 2e-5)
    check(f"{tag} dgam1", d1, lg1.grad[0], 2e-5)
    if both:
        check(f"{tag} gz_after", gza.cpu() - seed, lza.grad, 1e-5)
        check(f"{tag} dgam2", d2, lg2.grad[0], 2e-5)


# ---------------------------------------------------------------------------------- nets
def test_csr_chain_matches_reference():
    """traincsr.py:203-204: first frame without a neighbour (second bank), then two recurrent calls
    whose codes carry gradient from one call into the previous one."""
    g = load_golden("c1_csr_chain")
    net = build(g, "CDLNet_CSR")
    sig = dev(g["sigma"])
    y0, y1, x0, x1 = (g[k].cuda() for k in ("y0", "y1", "x0", "x1"))
    xh0, z0 = net(y0, None, sig)
    xh1, z1 = net(y1, z0, sig)
    xh0b, z0b = net(y0, z1, sig)
    for got, key in ((xh0, "xh0"), (z0, "z0"), (xh1, "xh1"), (z1, "z1"), (xh0b, "xh0b"), (z0b, "z0b")):
        check(f"c1 {key}", got, g[key], XTOL)
    mse = lambda a, b: torch.mean((a - b) ** 2)
    loss = mse(x0, xh0) + mse(x1, xh1) + mse(x0, xh0b)
    assert abs(loss.item() - g["loss"]) < 1e-6 * max(1.0, abs(g["loss"])) + 1e-8
    loss.backward()
    check_param_grads("c1", net, g)
    for key, ref, got in (("xh1", g["xh1"], xh1), ("xh0b", g["xh0b"], xh0b)):
        p_ref, p_got = O.psnr(g["x1" if key == "xh1" else "x0"], ref), O.psnr(g["x1" if key == "xh1" else "x0"], got.detach().cpu())
        log(f"c1 {key:56s} PSNR ref={p_ref:.4f} ours={p_got:.4f}")
        assert round(p_ref, 2) == round(p_got, 2)


def test_csr_stride2_odd_leaf_neighbour():
    g = load_golden("c1b_csr_s2_odd")
    net = build(g, "CDLNet_CSR")
    zprev = g["zprev"].cuda().requires_grad_(True)
    xh, z = net(g["y"].cuda(), zprev, g["sigma"])
    check("c1b xhat", xh, g["xhat"], XTOL)
    check("c1b z", z, g["z"], XTOL)
    loss = torch.mean((g["x"].cuda() - xh) ** 2) + 0.1 * z.abs().mean()
    assert abs(loss.item() - g["loss"]) < 1e-6 * max(1.0, abs(g["loss"])) + 1e-8
    loss.backward()
    check("c1b grad z_prev", zprev.grad, g["grad_zprev"], GTOL)
    check_param_grads("c1b", net, g)


def test_csr_inference_has_no_autograd_state():
    g = load_golden("c1b_csr_s2_odd")
    net = build(g, "CDLNet_CSR")
    with torch.no_grad():
        xh, z = net(g["y"].cuda(), g["zprev"].cuda(), g["sigma"])
    assert not xh.requires_grad and not z.requires_grad
    check("c1b no-grad xhat", xh, g["xhat"], XTOL)
    check("c1b no-grad z", z, g["z"], XTOL)


def test_csrf2_four_branches_match_reference():
    """traincsr.py:257-261: plain, previous-only, both neighbours, next-only, chained."""
    g = load_golden("c2_csrf2_chain")
    net = build(g, "CDLNet_CSRf2")
    s = g["sigma"]
    y = [g[f"y{i}"].cuda() for i in range(3)]
    x = [g[f"x{i}"].cuda() for i in range(3)]
    xp, zp = net(y[0], None, None, s)
    xc, zc = net(y[1], zp, None, s)
    xa, za = net(y[2], zc, None, s)
    xc2, zc2 = net(y[1], zp, za, s)
    xp2, zp2 = net(y[0], None, za, s)
    for got, key in ((xp, "xp"), (zp, "zp"), (xc, "xc"), (zc, "zc"), (xa, "xa"), (za, "za"),
                     (xc2, "xc2"), (zc2, "zc2"), (xp2, "xp2"), (zp2, "zp2")):
        check(f"c2 {key}", got, g[key], XTOL)
    mse = lambda a, b: torch.mean((a - b) ** 2)
    loss = mse(x[0], xp) + mse(x[1], xc) + mse(x[2], xa) + mse(x[1], xc2) + mse(x[0], xp2)
    assert abs(loss.item() - g["loss"]) < 1e-6 * max(1.0, abs(g["loss"])) + 1e-8
    loss.backward()
    check_param_grads("c2", net, g)


def test_csr_rejects_mismatched_neighbour():
    g = load_golden("c1b_csr_s2_odd")
    net = build(g, "CDLNet_CSR")
    with pytest.raises(ValueError):
        net(g["y"].cuda(), g["zprev"].cuda()[..., :-1], g["sigma"])


def test_csr_masked_frame_vs_oracle():
    """mask != 1 and a batch, larger than the fixtures, against the oracle (forward + every gradient)."""
    import cdlnet_video_amd as cva
    torch.manual_seed(5)
    K, M, P, s, C = 3, 8, 5, 1, 1
    net = cva.CDLNet_CSRf2(K=K, M=M, P=P, s=s, C=C, t0=5e-3, adaptive=True, init=True)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n in ("g1", "g2"):
                p.uniform_(0.2, 1.2)
            elif n == "t":
                p.uniform_(2e-3, 1.5e-2)
            else:
                p.add_(0.05 * p.abs().mean() * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = cva.utils.synthetic_clip((3, 1, 40, 36), seed=2)
    gen = torch.Generator().manual_seed(3)
    mask = (torch.rand(x.shape, generator=gen) > 0.3).float()
    sig = torch.tensor([10.0, 25.0, 40.0]).reshape(3, 1, 1, 1)
    y = mask * (x + torch.randn(x.shape, generator=gen) * sig / 255)
    kw = dict(K=K, P=P, s=s, sigma=sig, adaptive=True, mask=mask, variant="f2")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "D.weight"}
    leaves["D.weight"] = leaves["B.0.weight"]
    with torch.no_grad():
        _, zp = O.ista_csr(sd, y, None, None, **kw)
        _, za = O.ista_csr(sd, y.flip(0), None, None, **kw)
    xr, zr = O.ista_csr(leaves, y, zp, za, **kw)
    (torch.mean((x - xr) ** 2) + 0.05 * zr.abs().mean()).backward()
    xh, z = net(y.cuda(), zp.cuda(), za.cuda(), sig.cuda(), mask=mask.cuda())
    check("f2 masked xhat", xh, xr, XTOL)
    check("f2 masked z", z, zr, XTOL)
    (torch.mean((x.cuda() - xh) ** 2) + 0.05 * z.abs().mean()).backward()
    for pname, p in net.named_parameters():
        check(f"f2 masked grad {pname}", p.grad, leaves[pname].grad, GTOL)


# ---------------------------------------------------------------------------------- recurrent drivers
def _clip(T, shape, seed):
    import cdlnet_video_amd as cva
    vid = cva.utils.synthetic_clip((1, 1, T) + shape, seed=seed)           # (1,1,T,H,W): frames drift
    gen = torch.Generator().manual_seed(seed + 1)
    return [vid[:, :, t] + torch.randn(vid[:, :, t].shape, generator=gen) * 25 / 255 for t in range(T)], vid


def test_recurrent_loop_csr_vs_oracle():
    """analyzemri.py:87-156 call sequence; the oracle replays it with ista_csr."""
    import cdlnet_video_amd as cva
    g = load_golden("c1_csr_chain")
    net = build(g, "CDLNet_CSR")
    K, M, P, s, C = g["hyper"]
    frames, vid = _clip(5, (28, 36), 11)
    kw = dict(K=K, P=P, s=s, sigma=25.0, adaptive=True, variant="csr")
    with torch.no_grad():
        _, zp = O.ista_csr(g["sd"], frames[0], None, **kw)
        _, zc = O.ista_csr(g["sd"], frames[1], zp, **kw)
        first, zp = O.ista_csr(g["sd"], frames[0], zc, **kw)
        ref = [first]
        for t in range(1, 5):
            xh, zp = O.ista_csr(g["sd"], frames[t], zp, **kw)
            ref.append(xh)
    got = cva.csr_inference_loop(net, [f.cuda() for f in frames], 25.0)
    assert len(got) == 5
    for t in range(5):
        check(f"recurrent loop frame {t}", got[t], ref[t], XTOL)
        assert round(O.psnr(vid[:, :, t], ref[t]), 2) == round(O.psnr(vid[:, :, t], got[t].cpu()), 2)


def test_recurrent_v2_csrf2_vs_oracle():
    """analyzemri.py:162-182: causal pass recording codes, then the two-neighbour pass."""
    import cdlnet_video_amd as cva
    g = load_golden("c2_csrf2_chain")
    net = build(g, "CDLNet_CSRf2")
    K, M, P, s, C = g["hyper"]
    T = 4
    frames, vid = _clip(T, (26, 30), 17)
    kw = dict(K=K, P=P, s=s, sigma=25.0, adaptive=True, variant="f2")
    with torch.no_grad():
        codes = [None] * (T + 2)
        for t in range(T):
            _, codes[t + 1] = O.ista_csr(g["sd"], frames[t], codes[t], None, **kw)
        ref = [O.ista_csr(g["sd"], frames[t], codes[t], codes[t + 1], **kw)[0] for t in range(T)]
    got = cva.csr_inference_v2(net, [f.cuda() for f in frames], 25.0)
    for t in range(T):
        check(f"recurrent v2 frame {t}", got[t], ref[t], XTOL)


@pytest.mark.parametrize("both", [False, True])
@pytest.mark.parametrize("dims,P,s,C,M", [((24, 40), (5, 5), 1, 1, 11), ((22, 18), (7, 7), 2, 3, 9),
                                           ((4, 12, 16), (3, 5, 5), 1, 1, 6), ((20, 26), (11, 11), 1, 1, 5)])
def test_analysis_with_prox_epilogue_equals_two_calls(dims, P, s, C, M, both):
    """cdl_analysis_prox (tiled and untiled kernels) == cdl_analysis then cdl_prox_csr, bit for bit, and
    its u_out is the plain analysis output."""
    from cdlnet_video_amd import ops
    gen = torch.Generator().manual_seed(len(dims) * 7 + P[-1] + both)
    N = 2
    geom = ops.Geometry.make(N, C, M, dims, P, tuple(p // 2 for p in P), s)
    x = torch.randn(geom.image_shape(), generator=gen).cuda()
    w = (0.1 * torch.randn(geom.filter_shape(), generator=gen)).cuda()
    code = lambda: (0.3 * torch.randn(geom.code_shape(), generator=gen)
                    * (torch.rand(geom.code_shape(), generator=gen) > 0.5)).cuda()
    zin, zp, za = code(), code(), code()
    lam, g1, g2 = ((torch.rand(N, M, generator=gen) * sc).cuda() for sc in (0.2, 1.2, 1.2))
    u_ref = ops.analysis(geom, x, w, -1.0, zin, None, None)
    z_ref = ops.prox_csr(geom, u_ref, zp, lam, g1, za if both else None, g2 if both else None)
    u_out = torch.empty_like(u_ref)
    z = ops.analysis_prox(geom, x, w, -1.0, zin, zp, lam, g1, za if both else None, g2 if both else None, u_out=u_out)
    assert torch.equal(u_out, u_ref)
    assert torch.equal(z, z_ref)
    z2 = ops.analysis_prox(geom, x, w, -1.0, zin, zp, lam, g1, za if both else None, g2 if both else None)
    assert torch.equal(z2, z_ref)


# ---------------------------------------------------------------------------------- whole-sweep C calls
@pytest.mark.parametrize("dims,P,s,C,M,masked", [((24, 40), (5, 5), 1, 1, 11, False), ((22, 18), (7, 7), 2, 3, 9, True),
                                                  ((4, 12, 16), (3, 5, 5), 1, 1, 6, False)])
def test_generic_sweeps_equal_stepwise_launches(dims, P, s, C, M, masked):
    """cdl_ista_forward / cdl_ista_backward enqueue exactly the launches the stepwise Python loops make:
    every output is bit-identical, for the plain ST loop and for both CSR maps."""
    from cdlnet_video_amd import loop, ops
    gen = torch.Generator().manual_seed(sum(dims) + M)
    N, K = 2, 3
    geom = ops.Geometry.make(N, C, M, dims, P, tuple(p // 2 for p in P), s)
    rnd = lambda shape, sc=1.0: (sc * torch.randn(shape, generator=gen)).cuda()
    yp = rnd(geom.image_shape())
    mask = (torch.rand(geom.image_shape(), generator=gen) > 0.3).float().cuda() if masked else None
    A = [rnd(geom.filter_shape(), 0.08) for _ in range(K)]
    B = [rnd(geom.filter_shape(), 0.08) for _ in range(K)]
    c = torch.rand(N, generator=gen).cuda()
    t, g1, g2 = ((torch.rand(K, 2, M, 1, 1, generator=gen) * sc).cuda() for sc in (0.05, 1.0, 1.0))
    tau, gam1, gam2 = (ops.thresholds(p, c, N) for p in (t, g1, g2))
    g_xp, g_z = rnd(geom.image_shape()), rnd(geom.code_shape(), 0.1)

    # plain loop
    a = loop._forward_generic(geom, yp, mask, tau, A, B, True, True)
    b = loop._forward_generic_stepwise(geom, yp, mask, tau, A, B, True, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2])) and all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
    dta, dtb = torch.zeros(K, 2, M, device="cuda"), torch.zeros(K, 2, M, device="cuda")
    ga = loop._backward_generic(geom, K, yp, mask, c, A, B, a[2], a[3], g_xp, g_z, dta)
    gb = loop._backward_generic_stepwise(geom, K, yp, mask, c, A, B, b[2], b[3], g_xp, g_z, dtb)
    assert torch.equal(dta, dtb)
    for x, y in zip(ga[0] + ga[1], gb[0] + gb[1]):
        assert torch.equal(x, y)
    inf = loop._forward_generic(geom, yp, mask, tau, A, B, False, False)          # ping-pong buffers
    assert torch.equal(inf[0], a[0]) and torch.equal(inf[1], a[1])

    # CSR loops (one and two neighbours)
    zp, za = rnd(geom.code_shape(), 0.2), rnd(geom.code_shape(), 0.2)
    for zaft, gm2 in ((None, None), (za, gam2)):
        a = loop._forward_csr(geom, yp, mask, tau, gam1, gm2, zp, zaft, A, B, True)
        b = loop._forward_csr_stepwise(geom, yp, mask, tau, gam1, gm2, zp, zaft, A, B, True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for la, lb in zip(a[2:], b[2:]):
            assert len(la) == len(lb) and all(torch.equal(x, y) for x, y in zip(la, lb))
        inf = loop._forward_csr(geom, yp, mask, tau, gam1, gm2, zp, zaft, A, B, False)
        assert torch.equal(inf[0], a[0]) and torch.equal(inf[1], a[1])
