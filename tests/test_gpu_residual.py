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
    for key, ref in g["grad"].items():
        check(f"{name} grad {key}", params[key].grad, ref, 2e-4)
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
    """Channel counts and extents on both sides of the matrix-core tier's eligibility (M % 16, tile edges)."""
    gen = torch.Generator().manual_seed(100 + M)
    x = torch.randn((N, M) + shape, generator=gen) * 0.5
    w1 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    w2 = torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5
    wgt = torch.randn((N, M) + shape, generator=gen)
    xo, w1o, w2o = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    ref = orc.residual_block(xo, w1o, w2o)
    (ref * wgt).sum().backward()
    xg, w1g, w2g = (t.cuda().requires_grad_(True) for t in (x, w1, w2))
    out = cva.loop.ResidualBlockFn.apply(xg, w1g, w2g)
    check(f"block M={M} {shape} out", out, ref, 1e-5)
    (out * wgt.cuda()).sum().backward()
    check(f"block M={M} {shape} grad_x", xg.grad, xo.grad, 1e-4)
    check(f"block M={M} {shape} grad_w1", w1g.grad, w1o.grad, 1e-4)
    check(f"block M={M} {shape} grad_w2", w2g.grad, w2o.grad, 1e-4)
