"""GPU parity of ResidualBlock and CDLNetVideo(residual=True) (reference
model/net.py:105-227) against the reference-generated fixtures r0-r2 and, at larger sizes, the oracle.
Tolerance: 1e-5 relative on outputs (fp32 storage, split-bf16 x3 or fp32 arithmetic), 1e-4 on gradients."""
import pytest
import torch

import cdlnet_video_amd as cva
from gpu_util import check, load_golden
from oracle import cdl_oracle as orc

pytestmark = pytest.mark.gpu


def test_block_matches_reference_fixture():
    g = load_golden("r0_residual_block")
    blk = cva.ResidualBlock(8, 8)
    blk.load_state_dict(g["sd"])
    blk = blk.cuda()
    x = g["x"].cuda().requires_grad_(True)
    out = blk(x)
    check("r0 block out", out, g["out"], 1e-5)
    (out * g["weight"].cuda()).sum().backward()
    check("r0 block grad_x", x.grad, g["grad_x"], 1e-4)
    check("r0 block grad conv1", blk.conv1.weight.grad, g["grad"]["conv1.weight"], 1e-4)
    check("r0 block grad conv2", blk.conv2.weight.grad, g["grad"]["conv2.weight"], 1e-4)


@pytest.mark.parametrize("name,adaptive,code_loss", [("r1_video_residual", True, 0.05),
                                                     ("r2_video_residual_s2", False, 0.0)])
def test_video_residual_matches_reference_fixture(name, adaptive, code_loss):
    g = load_golden(name)
    K, M, Pd, Ph, Pw, s, C = g["hyper"]
    net = cva.CDLNetVideo(K=K, M=M, P=(Pd, Ph, Pw), s=s, C=C, t0=0.0, adaptive=adaptive, init=False, residual=True)
    assert set(net.state_dict().keys()) == set(g["sd"].keys())
    net.load_state_dict(g["sd"])
    net = net.cuda()
    sigma = g["sigma"].cuda() if torch.is_tensor(g["sigma"]) else g["sigma"]
    xhat, z = net(g["y"].cuda(), sigma)
    check(f"{name} xhat", xhat, g["xhat"], 1e-5)
    check(f"{name} z", z, g["z"], 1e-5)
    loss = torch.mean((g["x"].cuda() - xhat) ** 2)
    if code_loss:
        loss = loss + code_loss * z.abs().mean()
    assert abs(float(loss.detach()) - g["loss"]) < 1e-5 * max(1.0, abs(g["loss"]))
    loss.backward()
    params = dict(net.named_parameters())
    failures = []
    for key, ref in g["grad"].items():
        try:
            check(f"{name} grad {key}", params[key].grad, ref, 1e-4)
        except AssertionError as e:
            failures.append(str(e))
    assert not failures, failures
    # forward_generator: the ST outputs, then xhat
    with torch.no_grad():
        outs = list(net.forward_generator(g["y"].cuda(), sigma))
    assert len(outs) == K + 1
    sd = {k: v for k, v in g["sd"].items()}
    _, _, shrunk = orc.ista_video_residual(sd, g["y"], K=K, P=(Pd, Ph, Pw), s=s, sigma=g["sigma"], adaptive=adaptive,
                                           all_codes=True)
    for k in range(K):
        check(f"{name} generator code {k}", outs[k], shrunk[k], 1e-5)
    check(f"{name} generator xhat", outs[-1], g["xhat"], 1e-5)


@pytest.mark.parametrize("N,M,shape", [(1, 64, (6, 40, 72)), (2, 32, (3, 17, 33)), (1, 16, (4, 8, 8)),
                                       (1, 48, (5, 24, 36))])
def test_block_vs_oracle_dense_shapes(N, M, shape):
    """Channel counts and extents on both sides of the matrix-core tier's eligibility (M % 16, tile edges).
    Forward: 1e-5 on h and out, and the ReLU supports agree except where a pre-activation is within the
    split-bf16 error of 0 (a handful of ~1e5 elements).  Backward: the block is linear given its two gates, so the
    gradients are checked against autograd on the oracle's convolutions with the PRODUCT's gates -- a single
    flipped gate would otherwise move one element of grad_x by its full magnitude (max-norm 1e-1)."""
    import torch.nn.functional as F
    o = cva.ops
    gen = torch.Generator().manual_seed(100 + M)
    x = torch.randn((N, M) + shape, generator=gen) * 0.5
    w1 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    w2 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    wgt = torch.randn((N, M) + shape, generator=gen)
    geom = o.residual_geometry(x, w1)
    xd, w1d, w2d = x.cuda(), w1.cuda(), w2.cuda()
    h, out = o.residual_forward(geom, xd, w1d, w2d)
    ref_h = torch.relu(F.conv3d(x, w1, padding=1))
    ref = orc.residual_block(x, w1, w2)
    tag = f"block M={M} {shape}"
    check(tag + " h", h, ref_h, 1e-5)
    check(tag + " out", out, ref, 1e-5)
    G1, G2 = (h.cpu() != 0), (out.cpu() != 0)
    flips = int((G1 != (ref_h != 0)).sum() + (G2 != (ref != 0)).sum())
    assert flips <= max(2, x.numel() // 20000), flips
    xo, w1o, w2o = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    lin = (F.conv3d(F.conv3d(xo, w1o, padding=1) * G1, w2o, padding=1) + xo) * G2
    (lin * wgt).sum().backward()
    dx, dw1, dw2 = o.residual_backward(geom, xd, h, out, w1d, w2d, wgt.cuda())
    check(tag + " grad_x", dx, xo.grad, 2e-5)
    check(tag + " grad_w1", dw1, w1o.grad, 2e-5)
    check(tag + " grad_w2", dw2, w2o.grad, 2e-5)
    # the autograd node is the same two calls
    xg, w1g, w2g = (t.cuda().requires_grad_(True) for t in (x, w1, w2))
    out2 = cva.loop.ResidualBlockFn.apply(xg, w1g, w2g)
    (out2 * wgt.cuda()).sum().backward()
    assert torch.equal(out2.detach(), out) and torch.equal(xg.grad, dx) and torch.equal(w1g.grad, dw1)
