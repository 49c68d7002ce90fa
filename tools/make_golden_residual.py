#!/usr/bin/env python3
"""Generate tests/golden/r*.npz (CDLNetVideo with residual=True and the ResidualBlock on its own, SURVEY.md
section 8(f) item 4) by running the UNMODIFIED reference classes (model/net.py:105-227) on CPU.  Same shims and
rules as tools/make_golden.py; its own script so the existing fixtures (RNG call order) are untouched.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_residual.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, save, smooth, state, grads_of   # noqa: E402


class ReluMargins:
    """Forward hooks (observers only) on a reference ResidualBlock: the smallest non-zero |pre-activation| of its
    two ReLUs.  A pre-activation within rounding of 0 makes the ReLU gate -- and with it every gradient -- depend
    on summation order, so fixtures are drawn until the margin is far above fp32 rounding."""

    def __init__(self, blocks):
        self.margin = float("inf")
        for blk in blocks:
            blk.register_forward_pre_hook(self._keep_input)
            blk.conv1.register_forward_hook(self._conv1)
            blk.conv2.register_forward_hook(self._conv2)

    def _note(self, pre):
        a = pre.detach().abs()
        a = a[a > 0]
        if a.numel():
            self.margin = min(self.margin, float(a.min()))

    def _keep_input(self, mod, args):
        self.x = args[0].detach().clone()

    def _conv1(self, mod, args, out):
        self._note(out)

    def _conv2(self, mod, args, out):
        self._note(out.detach() + self.x)


def main():
    net_mod, _ = import_reference()
    g = torch.Generator().manual_seed(9876)
    mse = lambda a, b: torch.mean((a - b) ** 2)

    # ---- R0: one ResidualBlock, forward and the three gradients ------------------------------------
    torch.manual_seed(31)
    blk = net_mod.ResidualBlock(8, 8)
    x = (torch.randn((2, 8, 5, 9, 11), generator=g) * 0.5).requires_grad_(True)
    out = blk(x)
    wgt = torch.randn(out.shape, generator=g)
    (out * wgt).sum().backward()
    save("r0_residual_block", x=x, out=out, weight=wgt, grad_x=x.grad, **state(blk), **grads_of(blk))

    # ---- R1: CDLNetVideo(residual=True), adaptive thresholds, loss on xhat and on the code ---------
    for seed in range(32, 4032, 100):
        torch.manual_seed(seed)
        net = net_mod.CDLNetVideo(K=3, M=8, P=(3, 5, 5), s=1, C=1, t0=5e-3, adaptive=True, depth=3, init=True,
                                  residual=True)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n == "t":
                    p.copy_(torch.rand(p.shape, generator=g) * 1.3e-2 + 2e-3)
                else:
                    p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=g))
        xc = smooth((2, 1, 4, 12, 14), g)
        sig = torch.tensor([15.0, 30.0]).reshape(2, 1, 1, 1, 1)
        y = xc + torch.randn(xc.shape, generator=g) * sig / 255
        watch = ReluMargins(net.residual_blocks)
        xhat, z = net(y, sig)
        print(f"r1 seed {seed}: smallest non-zero ReLU pre-activation {watch.margin:.2e}")
        if watch.margin > 1e-6:
            break
    else:
        raise SystemExit("no seed with a safe ReLU margin")
    loss = mse(xc, xhat) + 0.05 * z.abs().mean()
    loss.backward()
    save("r1_video_residual", x=xc, y=y, sigma=sig, xhat=xhat, z=z, loss=loss, **state(net), **grads_of(net),
         hyper=np.array([3, 8, 3, 5, 5, 1, 1]))

    # ---- R2: stride 2, odd extents (stride padding), constant sigma, no code loss ------------------
    for seed in range(33, 1033, 100):
        torch.manual_seed(seed)
        net = net_mod.CDLNetVideo(K=2, M=16, P=(3, 5, 5), s=2, C=1, t0=1e-2, adaptive=False, depth=3, init=True,
                                  residual=True)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n == "t":
                    p.copy_(torch.rand(p.shape, generator=g) * 1.3e-2 + 2e-3)
                else:
                    p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=g))
        xc = smooth((1, 1, 7, 21, 19), g)
        y = xc + torch.randn(xc.shape, generator=g) * 25 / 255
        watch = ReluMargins(net.residual_blocks)
        xhat, z = net(y, 25.0)
        print(f"r2 seed {seed}: smallest non-zero ReLU pre-activation {watch.margin:.2e}")
        if watch.margin > 2e-6:
            break
    else:
        raise SystemExit("no seed with a safe ReLU margin")
    loss = mse(xc, xhat)
    loss.backward()
    save("r2_video_residual_s2", x=xc, y=y, sigma=25.0, xhat=xhat, z=z, loss=loss, **state(net), **grads_of(net),
         hyper=np.array([2, 16, 3, 5, 5, 2, 1]))


if __name__ == "__main__":
    main()
