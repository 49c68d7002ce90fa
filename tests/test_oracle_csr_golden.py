"""The CSR temporal variants of the CPU oracle (oracle/cdl_oracle.py: prox_csr, prox_csr_f2, ista_csr)
replayed against fixtures generated from the unmodified reference classes CDLNet_CSR / CDLNet_CSRf2
(tools/make_golden_csr.py).  Pins the oracle for SURVEY.md section 8(f) item 1."""
import torch

from conftest import load_golden, rel_err
from oracle import cdl_oracle as O

TOL = 2e-6


def leaf_state(g):
    return {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}


def check_grads(sd, g, tol=5e-5):
    n = 0
    for key, ref in g["grad"].items():
        got = sd[key].grad
        if got is None:                      # parameter unused by every call of the chain
            assert float(ref.abs().max()) == 0.0, key
            continue
        assert rel_err(got, ref) < tol, key
        n += 1
    return n


def test_prox_pointwise():
    g = load_golden("c0_prox_pointwise")
    cases = g["cases"].reshape(-1, 3)
    for i, (lam, g1, g2) in enumerate(cases.tolist()):
        p1 = O.prox_csr(g["u"], g["zp"], torch.tensor(lam), torch.tensor(g1))
        p2 = O.prox_csr_f2(g["u"], g["zp"], g["za"], torch.tensor(lam), torch.tensor(g1), torch.tensor(g2))
        assert torch.equal(p1, g["prox_csr"][i])
        assert torch.equal(p2, g["prox_csr_f2"][i])


def test_csr_chain_forward_and_grads():
    g = load_golden("c1_csr_chain")
    K, M, P, s, C = g["hyper"]
    sd = leaf_state(g)
    kw = dict(K=K, P=P, s=s, sigma=g["sigma"], adaptive=True, variant="csr")
    xh0, z0 = O.ista_csr(sd, g["y0"], None, **kw)
    xh1, z1 = O.ista_csr(sd, g["y1"], z0, **kw)
    xh0b, z0b = O.ista_csr(sd, g["y0"], z1, **kw)
    for got, key in ((xh0, "xh0"), (z0, "z0"), (xh1, "xh1"), (z1, "z1"), (xh0b, "xh0b"), (z0b, "z0b")):
        assert rel_err(got, g[key]) < TOL, key
    mse = lambda a, b: torch.mean((a - b) ** 2)
    loss = mse(g["x0"], xh0) + mse(g["x1"], xh1) + mse(g["x0"], xh0b)
    assert abs(float(loss.detach()) - g["loss"]) < 1e-7
    loss.backward()
    assert check_grads(sd, g) >= 14          # A, B, A2 for every k, B2 for k >= 1 (D is B[0]), t, t2, g


def test_csr_stride2_leaf_neighbour():
    g = load_golden("c1b_csr_s2_odd")
    K, M, P, s, C = g["hyper"]
    sd = leaf_state(g)
    zprev = g["zprev"].clone().requires_grad_(True)
    xh, z = O.ista_csr(sd, g["y"], zprev, K=K, P=P, s=s, sigma=g["sigma"], adaptive=True, variant="csr")
    assert rel_err(xh, g["xhat"]) < TOL and rel_err(z, g["z"]) < TOL
    loss = torch.mean((g["x"] - xh) ** 2) + 0.1 * z.abs().mean()
    assert abs(float(loss.detach()) - g["loss"]) < 1e-7
    loss.backward()
    assert rel_err(zprev.grad, g["grad_zprev"]) < 5e-5
    assert check_grads(sd, g) >= 5
    assert sd["A2.0.weight"].grad is None    # the neighbour branch never touches the second bank


def test_csrf2_chain_forward_and_grads():
    g = load_golden("c2_csrf2_chain")
    K, M, P, s, C = g["hyper"]
    sd = leaf_state(g)
    kw = dict(K=K, P=P, s=s, sigma=g["sigma"], adaptive=True, variant="f2")
    xp, zp = O.ista_csr(sd, g["y0"], None, None, **kw)
    xc, zc = O.ista_csr(sd, g["y1"], zp, None, **kw)
    xa, za = O.ista_csr(sd, g["y2"], zc, None, **kw)
    xc2, zc2 = O.ista_csr(sd, g["y1"], zp, za, **kw)
    xp2, zp2 = O.ista_csr(sd, g["y0"], None, za, **kw)
    for got, key in ((xp, "xp"), (zp, "zp"), (xc, "xc"), (zc, "zc"), (xa, "xa"), (za, "za"),
                     (xc2, "xc2"), (zc2, "zc2"), (xp2, "xp2"), (zp2, "zp2")):
        assert rel_err(got, g[key]) < TOL, key
    mse = lambda a, b: torch.mean((a - b) ** 2)
    loss = mse(g["x0"], xp) + mse(g["x1"], xc) + mse(g["x2"], xa) + mse(g["x1"], xc2) + mse(g["x0"], xp2)
    assert abs(float(loss.detach()) - g["loss"]) < 1e-7
    loss.backward()
    assert check_grads(sd, g) >= 9
