#!/usr/bin/env python3
"""Generate tests/golden/c*.npz (the CSR temporal variants) by running
the UNMODIFIED reference classes CDLNet_CSR / CDLNet_CSRf2 and prox_CSR / prox_CSR_f2
(model/net.py:229-262, 363-463, 464-568) on CPU.  Same shims and rules as tools/make_golden.py;
kept in its own script so the existing fixtures (whose values depend on RNG call order) are untouched.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_csr.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, save, smooth, state, grads_of   # noqa: E402


def detie(net, gen, names_lo_hi):
    """De-tie the K copies of W; give every threshold family its own random values."""
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n in names_lo_hi:
                lo, hi = names_lo_hi[n]
                p.copy_(torch.rand(p.shape, generator=gen) * (hi - lo) + lo)
            else:
                p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=gen))


def main():
    net_mod, _ = import_reference()
    g = torch.Generator().manual_seed(4321)
    mse = lambda a, b: torch.mean((a - b) ** 2)

    # ---- C0: the two proximal maps pointwise, incl. exact zeros, ties and negative thresholds ----
    grid = torch.linspace(-0.06, 0.06, 13)
    u, zp, za = torch.meshgrid(grid, grid[::2], grid[::3], indexing="ij")
    u, zp, za = u.reshape(-1), zp.reshape(-1), za.reshape(-1)
    cases = [(0.01, 0.5, 0.8), (0.02, 1.5, 0.3), (0.0, 1.0, 1.0), (0.01, -0.5, 0.7), (-0.005, 0.6, -0.4)]
    p1 = torch.stack([net_mod.prox_CSR(u, zp, torch.tensor(l), torch.tensor(g1)) for l, g1, _ in cases])
    p2 = torch.stack([net_mod.prox_CSR_f2(u, zp, za, torch.tensor(l), torch.tensor(g1), torch.tensor(g2))
                      for l, g1, g2 in cases])
    save("c0_prox_pointwise", u=u, zp=zp, za=za, cases=np.array(cases), prox_csr=p1, prox_csr_f2=p2)

    # ---- C1: CDLNet_CSR, two-frame chain of traincsr.py:203-204 (first frame on A2/B2/t2) --------
    torch.manual_seed(21)
    net = net_mod.CDLNet_CSR(K=3, M=6, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    detie(net, g, {"t": (2e-3, 1.5e-2), "t2": (2e-3, 1.5e-2), "g": (0.2, 1.4)})
    x = smooth((2, 1, 20, 24), g)
    x1 = (0.9 * x + 0.1 * smooth((2, 1, 20, 24), g)).clamp(0, 1)
    sig = torch.tensor([15.0, 30.0]).reshape(2, 1, 1, 1)
    y0 = x + torch.randn(x.shape, generator=g) * sig / 255
    y1 = x1 + torch.randn(x.shape, generator=g) * sig / 255
    xh0, z0 = net(y0, None, sig)
    xh1, z1 = net(y1, z0, sig)
    xh0b, z0b = net(y0, z1, sig)                  # traincsr.py:203: previous frame re-estimated from z_curr
    loss = mse(x, xh0) + mse(x1, xh1) + mse(x, xh0b)
    loss.backward()
    save("c1_csr_chain", x0=x, x1=x1, y0=y0, y1=y1, sigma=sig, xh0=xh0, z0=z0, xh1=xh1, z1=z1, xh0b=xh0b,
         z0b=z0b, loss=loss, **state(net), **grads_of(net), hyper=np.array([3, 6, 5, 1, 1]))

    # ---- C1b: CDLNet_CSR, stride 2, odd size, float sigma, gradient w.r.t. a leaf z_prev ----------
    torch.manual_seed(22)
    net = net_mod.CDLNet_CSR(K=2, M=5, P=7, s=2, C=1, t0=1e-2, adaptive=True, init=True)
    detie(net, g, {"t": (2e-3, 2e-2), "t2": (2e-3, 2e-2), "g": (0.3, 1.2)})
    x = smooth((1, 1, 19, 21), g)
    y = x + torch.randn(x.shape, generator=g) * 25 / 255
    with torch.no_grad():
        _, zseed = net(y, None, 25.0)
    zprev = (zseed + 0.01 * torch.randn(zseed.shape, generator=g) * (zseed != 0)).requires_grad_(True)
    xh, z = net(y, zprev, 25.0)
    loss = mse(x, xh) + 0.1 * z.abs().mean()      # the returned code carries gradient too
    loss.backward()
    save("c1b_csr_s2_odd", x=x, y=y, sigma=25.0, zprev=zprev, xhat=xh, z=z, loss=loss,
         grad_zprev=zprev.grad, **state(net), **grads_of(net), hyper=np.array([2, 5, 7, 2, 1]))

    # ---- C2: CDLNet_CSRf2, the four branches chained as in traincsr.py:257-261 --------------------
    torch.manual_seed(23)
    net = net_mod.CDLNet_CSRf2(K=3, M=6, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    detie(net, g, {"t": (2e-3, 1.5e-2), "g1": (0.2, 1.4), "g2": (0.2, 1.4)})
    xs = [smooth((1, 1, 18, 22), g)]
    for _ in range(2):
        xs.append((0.9 * xs[-1] + 0.1 * smooth((1, 1, 18, 22), g)).clamp(0, 1))
    ys = [xx + torch.randn(xx.shape, generator=g) * 25 / 255 for xx in xs]
    xp_, zp_ = net(ys[0], None, None, 25.0)                # plain ISTA branch
    xc_, zc_ = net(ys[1], zp_, None, 25.0)                 # previous only  (g1)
    xa_, za_ = net(ys[2], zc_, None, 25.0)
    xc2, zc2 = net(ys[1], zp_, za_, 25.0)                  # both neighbours (prox_CSR_f2)
    xp2, zp2 = net(ys[0], None, za_, 25.0)                 # next only      (g2)
    loss = mse(xs[0], xp_) + mse(xs[1], xc_) + mse(xs[2], xa_) + mse(xs[1], xc2) + mse(xs[0], xp2)
    loss.backward()
    save("c2_csrf2_chain", x0=xs[0], x1=xs[1], x2=xs[2], y0=ys[0], y1=ys[1], y2=ys[2], sigma=25.0,
         xp=xp_, zp=zp_, xc=xc_, zc=zc_, xa=xa_, za=za_, xc2=xc2, zc2=zc2, xp2=xp2, zp2=zp2, loss=loss,
         **state(net), **grads_of(net), hyper=np.array([3, 6, 5, 1, 1]))


if __name__ == "__main__":
    main()
