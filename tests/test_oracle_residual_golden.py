"""CPU: the oracle's restatement of ResidualBlock / CDLNetVideo(residual=True) (model/net.py:105-227) against the
fixtures tools/make_golden_residual.py generated from the unmodified reference."""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cdl_oracle as orc


def test_residual_block_matches_reference():
    g = load_golden("r0_residual_block")
    x = g["x"].clone().requires_grad_(True)
    w1 = g["sd"]["conv1.weight"].clone().requires_grad_(True)
    w2 = g["sd"]["conv2.weight"].clone().requires_grad_(True)
    out = orc.residual_block(x, w1, w2)
    assert rel_err(out, g["out"]) < 1e-6
    (out * g["weight"]).sum().backward()
    assert rel_err(x.grad, g["grad_x"]) < 1e-5
    assert rel_err(w1.grad, g["grad"]["conv1.weight"]) < 1e-5
    assert rel_err(w2.grad, g["grad"]["conv2.weight"]) < 1e-5


@pytest.mark.parametrize("name,adaptive,code_loss", [("r1_video_residual", True, 0.05),
                                                     ("r2_video_residual_s2", False, 0.0)])
def test_video_residual_matches_reference(name, adaptive, code_loss):
    g = load_golden(name)
    K, M, Pd, Ph, Pw, s, C = g["hyper"]
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    xhat, z = orc.ista_video_residual(sd, g["y"], K=K, P=(Pd, Ph, Pw), s=s, sigma=g["sigma"], adaptive=adaptive)
    assert rel_err(xhat, g["xhat"]) < 1e-5
    assert rel_err(z, g["z"]) < 1e-5
    loss = torch.mean((g["x"] - xhat) ** 2)
    if code_loss:
        loss = loss + code_loss * z.abs().mean()
    assert abs(float(loss.detach()) - g["loss"]) < 1e-6 * max(1.0, abs(g["loss"]))
    loss.backward()
    for key, ref in g["grad"].items():
        assert rel_err(sd[key].grad, ref) < 1e-4, key
