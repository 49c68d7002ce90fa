"""Plain-C direct-loop oracle vs the PyTorch restatement (which is pinned to the reference)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import c_oracle, cdl_oracle as O


@pytest.mark.parametrize("name,ndim", [("f1_2d_s1", 2), ("f2_2d_s2_odd", 2), ("f3b_jdd_s2_odd", 2),
                                       ("f4c_3d_s2_odd", 3), ("f6_negative_t", 2)])
def test_c_loop_matches_torch_restatement(name, ndim):
    g = load_golden(name)
    K, M, P, s, C = g["hyper"]
    P = tuple(g["P3"]) if "P3" in g else P
    sd = g["sd"]
    yp, mean, pads, mask_p = O.preprocess(g["y"], s, g.get("mask"))
    c = g["sigma"] / 255.0
    N = yp.shape[0]
    tau = torch.stack([(sd["t"][k, 0] + c * sd["t"][k, 1]).reshape(-1, M).expand(N, M) for k in range(K)])
    wA = torch.stack([sd[f"A.{k}.weight"] for k in range(K)])
    wB = torch.stack([sd[f"B.{k}.weight"] for k in range(K)])
    z, xp = c_oracle.forward(yp.numpy(), None if mask_p is None else mask_p.numpy(), wA.numpy(),
                             wB.numpy(), tau.numpy(), O._conv_pad(P, ndim), s)
    xhat = O.postprocess(torch.from_numpy(xp), mean, pads)
    assert rel_err(xhat, g["xhat"]) < 2e-6
    assert rel_err(torch.from_numpy(z), g["z"] if "z" in g else g[f"code{K-1}"]) < 2e-6
