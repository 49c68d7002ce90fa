"""Net-level parity on the GPU: the product modules (HIP kernels through the C ABI) against the
reference's own outputs (tests/golden, generated from the unmodified reference) and against the
CPU oracle on larger seeded inputs.  Gate: max|xhat-ref|/max|ref| <= 1e-5 (BASELINE north_star)."""
import math

import pytest
import torch

from gpu_util import build_from_golden, check, hyper, load_golden, log
from oracle import cdl_oracle as O

pytestmark = pytest.mark.gpu

XTOL = 1e-5          # the north_star gate on xhat
ZTOL = 1e-5
GTOL = 2e-4          # parameter gradients (sums over every pixel of products of fp32 values)

CASES = [("f1_2d_s1", "2d"), ("f2_2d_s2_odd", "2d"), ("f3_jdd_c3_mask", "2d"), ("f3b_jdd_s2_odd", "2d"),
         ("f4a_3d_p555", "3d"), ("f4b_3d_p995_s2", "3d"), ("f4c_3d_s2_odd", "3d"),
         ("f5_gabor_shared", "gabor"), ("f5b_gabor_plain", "gabor"), ("f6_negative_t", "2d"),
         # reference outputs on geometries the fused MFMA kernels take (tools/make_golden_fused.py)
         ("f10_fused_m32_p7", "2d"), ("f11_fused_m64_p5", "2d")]


def run_case(name, kind):
    g = load_golden(name)
    extra = {"adaptive": name != "f5b_gabor_plain"} if kind == "gabor" else {}
    net = build_from_golden(g, kind, **extra)
    sigma = g["sigma"]
    if torch.is_tensor(sigma):
        sigma = sigma.cuda()
    mask = g["mask"].cuda() if "mask" in g else 1
    return g, net, sigma, mask


@pytest.mark.parametrize("name,kind", CASES)
def test_forward_matches_reference(name, kind):
    g, net, sigma, mask = run_case(name, kind)
    with torch.no_grad():
        xhat, z = net(g["y"].cuda(), sigma, mask=mask)
    assert xhat.shape == g["xhat"].shape
    check(f"{name} xhat", xhat, g["xhat"], XTOL)
    ref_z = g["z"] if "z" in g else g[f"code{hyper(g)[0] - 1}"]
    check(f"{name} z_K", z, ref_z, ZTOL)
    p_ref, p_got = O.psnr(g["x"], g["xhat"]), O.psnr(g["x"], xhat.cpu())
    log(f"{name:60s} PSNR ref={p_ref:.4f} ours={p_got:.4f}")
    assert round(p_ref, 2) == round(p_got, 2)


@pytest.mark.parametrize("name,kind", CASES)
def test_gradients_match_reference(name, kind):
    g, net, sigma, mask = run_case(name, kind)
    xhat, _ = net(g["y"].cuda(), sigma, mask=mask)
    loss = torch.mean((g["x"].cuda() - xhat) ** 2)
    loss.backward()
    assert abs(loss.item() - g["loss"]) < 1e-6 * max(1.0, abs(g["loss"])) + 1e-8
    seen = 0
    for pname, p in net.named_parameters():
        if pname == "g":
            assert p.grad is None
            continue
        ref = g["grad"][pname]
        assert p.grad is not None, pname
        check(f"{name} grad {pname}", p.grad, ref, GTOL)
        seen += 1
    assert seen == len(g["grad"])


@pytest.mark.parametrize("name,kind", [("f1_2d_s1", "2d"), ("f4a_3d_p555", "3d")])
def test_forward_generator_codes(name, kind):
    g, net, sigma, mask = run_case(name, kind)
    K = hyper(g)[0]
    with torch.no_grad():
        items = list(net.forward_generator(g["y"].cuda(), sigma, mask=mask))
    assert len(items) == K + 1
    for k in range(K):
        check(f"{name} generator code {k}", items[k], g[f"code{k}"], ZTOL)
    check(f"{name} generator xhat", items[-1], g["xhat"], XTOL)


def test_z_output_gradient_flows():
    """A loss on the code z_K (not only on xhat) back-propagates; oracle autograd is the reference."""
    g, net, sigma, mask = run_case("f2_2d_s2_odd", "2d")
    xhat, z = net(g["y"].cuda(), sigma, mask=mask)
    (xhat.square().mean() + 0.1 * z.abs().mean()).backward()
    K, M, P, s, C = hyper(g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items() if k not in ("g", "D.weight")}
    leaves["D.weight"] = leaves["B.0.weight"]
    xr, zr = O.ista(leaves, g["y"], K=K, P=P, s=s, sigma=g["sigma"], adaptive=True)
    (xr.square().mean() + 0.1 * zr.abs().mean()).backward()
    for pname, p in net.named_parameters():
        if pname != "g":
            check(f"z-loss grad {pname}", p.grad, leaves[pname].grad, GTOL)


@pytest.mark.parametrize("name,kind", [("f7_train_step", "2d"), ("f7b_train_step_3d", "3d")])
def test_training_step_matches_reference(name, kind):
    """loss, every gradient, and the post-(clip, Adam, project) parameters of one reference step."""
    import cdlnet_video_amd as cva
    g = load_golden(name)
    net = build_from_golden(g, kind)
    opt = torch.optim.Adam(net.parameters(), lr=g["lr"])
    opt.zero_grad()
    xhat, _ = net(g["y"].cuda(), g["sigma"].cuda(), mask=1)
    loss = torch.mean((g["x"].cuda() - xhat) ** 2)
    loss.backward()
    assert abs(loss.item() - g["loss"]) < 1e-7
    for pname, p in net.named_parameters():
        if pname != "g":
            check(f"{name} grad {pname}", p.grad, g["grad"][pname], GTOL)
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), g["clip"])
    assert abs(total.item() - g["grad_norm"]) < 1e-4 * g["grad_norm"]
    opt.step()
    if kind == "2d":
        net.project()                          # train3d.py never projects
    sd = net.state_dict()
    for key, ref in g["after"].items():
        check(f"{name} after-step {key}", sd[key], ref, 2e-5)
    assert isinstance(net, (cva.CDLNet, cva.CDLNetVideo))


# ---- larger seeded cases against the CPU oracle --------------------------------------------
def oracle_case(kind, K, M, P, s, C, shape, seed, t0=5e-3, mask=False, sigma=25.0):
    import cdlnet_video_amd as cva
    torch.manual_seed(seed)
    if kind == "2d":
        net = cva.CDLNet(K=K, M=M, P=P, s=s, C=C, t0=t0, adaptive=True, init=True)
    else:
        net = cva.CDLNetVideo(K=K, M=M, P=P, s=s, C=C, t0=t0, adaptive=True, depth=shape[2], init=True)
    with torch.no_grad():                       # de-tie the K copies
        for n_, p in net.named_parameters():
            if n_ not in ("t", "g"):
                p.add_(0.03 * p.abs().mean() * torch.randn_like(p))
    gen = torch.Generator().manual_seed(seed + 100)
    x = cva.utils.synthetic_clip(shape, seed=seed)
    m = cva.gen_bayer_mask(x) if mask else None
    y = x + torch.randn(x.shape, generator=gen) * sigma / 255
    if m is not None:
        y = m * y
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.cuda(), sd, x, y, m


BIG = [
    ("cfg1 CDLNet K10 M32 P5 128x128", "2d", dict(K=10, M=32, P=5, s=1, C=1, shape=(1, 1, 128, 128), seed=1)),
    ("cfg2-arch CDLNet K30 M64 P7 2x96x96", "2d", dict(K=30, M=64, P=7, s=1, C=1, shape=(2, 1, 96, 96), seed=2)),
    ("s2030-arch K30 M169 P7 s2 75x77", "2d", dict(K=30, M=169, P=7, s=2, C=1, shape=(1, 1, 75, 77), seed=3)),
    ("cfg4-arch JDD C3 K12 M64 P7 64x64", "2d", dict(K=12, M=64, P=7, s=1, C=3, shape=(1, 3, 64, 64), seed=4, mask=True, sigma=10.0)),
    ("cfg3-arch Video K6 M48 P555 8x48x48", "3d", dict(K=6, M=48, P=[5, 5, 5], s=1, C=1, shape=(1, 1, 8, 48, 48), seed=5)),
]


@pytest.mark.parametrize("label,kind,kw", BIG)
def test_forward_vs_oracle_at_baseline_architectures(label, kind, kw):
    net, sd, x, y, m = oracle_case(kind, **kw)
    sigma = kw.get("sigma", 25.0)
    with torch.no_grad():
        xhat, z = net(y.cuda(), sigma, mask=1 if m is None else m.cuda())
    xr, zr = O.ista(sd, y, K=kw["K"], P=kw["P"], s=kw["s"], sigma=sigma, adaptive=True, mask=m,
                    ndim=2 if kind == "2d" else 3)
    check(f"{label} xhat", xhat, xr, XTOL)
    check(f"{label} z_K", z, zr, 2e-5)
    p_ref, p_got = O.psnr(x, xr), O.psnr(x, xhat.cpu())
    nnz = float((zr != 0).float().mean())
    log(f"{label:60s} PSNR ref={p_ref:.4f} ours={p_got:.4f} noisy={O.psnr(x, y if m is None else xr*0+y):.2f} nnz={nnz:.3f}")
    assert round(p_ref, 2) == round(p_got, 2)
    assert 0.005 < nnz < 0.9                    # both shrinkage branches are exercised


def test_grads_vs_oracle_mid_size():
    kw = dict(K=8, M=32, P=7, s=1, C=1, shape=(2, 1, 64, 64), seed=7)
    net, sd, x, y, m = oracle_case("2d", **kw)
    sig = torch.tensor([20.0, 30.0]).reshape(2, 1, 1, 1)
    xhat, _ = net(y.cuda(), sig.cuda())
    loss = torch.mean((x.cuda() - xhat) ** 2)
    loss.backward()
    sd["D.weight"] = sd["B.0.weight"]
    lref, grads, _ = O.loss_and_grads(sd, x, y, K=8, P=7, s=1, sigma=sig, adaptive=True)
    assert abs(loss.item() - lref) < 1e-6 * lref + 1e-9
    # 3e-3: ST support flips between two fp32 evaluations of the same net (see test_gpu_fused.py NET_GTOL)
    for pname, p in net.named_parameters():
        if pname != "g":
            check(f"mid-size grad {pname}", p.grad, grads[pname], 3e-3)


# ---- size-independent properties at a BASELINE-sized input -------------------------------------
def test_properties_at_full_cfg1_batch():
    """cfg1 at batch 10 (the reference's own loader batch): (i) samples are independent, so the
    batched result equals per-sample results bit for bit; (ii) ST support: z is exactly zero where
    the oracle's is zero-margin-safe; (iii) the output is finite and the mean is restored."""
    import cdlnet_video_amd as cva
    torch.manual_seed(11)
    net = cva.CDLNet(K=10, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=True).cuda()
    x = cva.utils.synthetic_clip((10, 1, 128, 128), seed=11)
    y, sig = cva.awgn(x, (20, 30), torch.Generator().manual_seed(12))
    with torch.no_grad():
        xhat, z = net(y.cuda(), sig.cuda())
        for n in (0, 7):
            xn, zn = net(y[n:n + 1].cuda(), sig[n:n + 1].cuda())
            assert torch.equal(xn, xhat[n:n + 1]) and torch.equal(zn, z[n:n + 1])
    assert torch.isfinite(xhat).all()
    assert abs(float(xhat.mean() - y.mean())) < 5e-3
    assert O.psnr(x, xhat.cpu()) > O.psnr(x, y)


def test_mcsure_objective_vs_oracle():
    """train.py:87-93: two forward passes share the parameters; loss and every gradient against autograd
    of the oracle with the same probe b.  (The reference has this inline in fit(): restated, not a fixture.)"""
    import cdlnet_video_amd as cva
    g, net, sigma, mask = run_case("f2_2d_s2_odd", "2d")
    K, M, P, s, C = hyper(g)
    y = g["y"]
    b = torch.randn(y.shape, generator=torch.Generator().manual_seed(12))
    ref_loss, ref_grads = O.mcsure_loss_and_grads(g["sd"], y, b, K=K, P=P, s=s, sigma=g["sigma"], adaptive=True)
    yd = y.cuda()
    xhat, _ = net(yd, sigma)
    loss = cva.mcsure_loss(net, yd, xhat, sigma, b=b.cuda())
    loss.backward()
    assert abs(loss.item() - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    for pname, p in net.named_parameters():
        if pname != "g":
            # the divergence term is a difference of two forwards divided by h = 1e-3: fp32 noise is amplified 1000x
            check(f"mcsure grad {pname}", p.grad, ref_grads[pname], 1e-3)


@pytest.mark.parametrize("precision", ["split3", "split4", "fp32"])
def test_mcsure_gradients_on_the_fused_path_with_both_supports_prescribed(precision):
    """the per-step MC-SURE gradient of the fused 2-D path (the f12 trajectory's geometry: K=3, M=32,
    P=5) against autograd of the oracle with the supports of BOTH forward passes prescribed (no support flip can enter;
    what remains is arithmetic, multiplied by 1/h = 1e3).  Gate 1e-4 of each tensor's maximum (measured: 6.7e-5 / 5.6e-5).  "split4" (all four bf16
    products) is measured beside the default: the two-term operand split (16-17 significant bits), not the dropped
    lo * lo product, is the floor, so it buys nothing.  "fp32": the same objective on the fp32 VALU
    tier (no fused kernel, no matrix cores)."""
    import cdlnet_video_amd as cva
    from cdlnet_video_amd import loop
    torch.manual_seed(31)
    K, M, P = 3, 32, 5
    net = cva.CDLNet(K=K, M=M, P=P, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_ == "t":
                p_.uniform_(2e-3, 2e-2)
            elif n_ != "g":
                p_.add_(0.05 * p_.abs().mean() * torch.randn_like(p_))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = cva.utils.synthetic_clip((2, 1, 48, 64), seed=9)
    y = x + torch.randn(x.shape, generator=torch.Generator().manual_seed(10)) * 25 / 255
    b = torch.randn(y.shape, generator=torch.Generator().manual_seed(12))
    h = 1e-3
    with loop.precision_scope(precision):
        outs = net._run(y.cuda(), 25.0, 1, True)
        outs_b = net._run((y + h * b).cuda(), 25.0, 1, True)
        s2 = (25.0 / 255.0) ** 2
        loss = torch.mean((y.cuda() - outs[0]) ** 2) + 2.0 * torch.mean(s2 * b.cuda() * (outs_b[0] - outs[0])) / h
        loss.backward()
    sup = [c.detach().cpu() for c in outs[2:]] + [outs[1].detach().cpu()]
    sup_b = [c.detach().cpu() for c in outs_b[2:]] + [outs_b[1].detach().cpu()]
    ref_loss, ref = O.mcsure_loss_and_grads(sd, y, b, K=K, P=P, s=1, sigma=25.0, adaptive=True, h=h, supports=sup,
                                            supports_b=sup_b)
    assert abs(loss.item() - ref_loss) < 2e-4 * max(1.0, abs(ref_loss))
    for pname, p_ in net.named_parameters():
        if pname != "g":
            check(f"mcsure[{precision}] K3 M32 P5 grad {pname} (both supports prescribed)", p_.grad, ref[pname],
                  2e-5 if precision == "fp32" else 1e-4)          # measured: fp32 6.3e-6, split3 6.7e-5, split4 5.6e-5


def test_fp32_scope_is_the_valu_tier_forward_and_backward(hip_env):
    """`loop.precision_scope("fp32")` = no fused kernel and no matrix cores: bit-identical (output and every gradient) to
    the generic backend with all four CDL_MFMA_* switches off, and within the 1e-5 gate of the default fused path.  The
    backward runs OUTSIDE the scope: the arithmetic is the forward's."""
    import cdlnet_video_amd as cva
    from cdlnet_video_amd import loop
    torch.manual_seed(77)
    net = cva.CDLNet(K=4, M=64, P=7, s=1, C=1, t0=5e-3, adaptive=True, init=True).cuda()
    x = cva.utils.synthetic_clip((2, 1, 64, 96), seed=5).cuda()
    y = x + torch.randn(x.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(6)) * 25 / 255

    def run():
        net.zero_grad(set_to_none=True)
        xhat, _ = net(y, 25.0)
        return xhat, torch.mean((x - xhat) ** 2)

    xd, ld = run()                                        # default: fused matrix-core sweep
    ld.backward()
    with loop.precision_scope("fp32"):
        xa, la = run()
    la.backward()                                         # outside the scope
    ga = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for name in ("CDL_MFMA_ANALYSIS", "CDL_MFMA_SYNTHESIS", "CDL_MFMA_WGRAD", "CDL_MFMA_DENSE"):
        hip_env(name, "0")
    loop.set_backend("generic")
    try:
        xb, lb = run()
        lb.backward()
    finally:
        loop.set_backend("auto")
    assert torch.equal(xa, xb)
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(ga[n], p.grad), n
    assert not torch.equal(xa, xd)
    check("fp32 tier vs fused sweep", xa, xd.cpu(), 1e-5)
