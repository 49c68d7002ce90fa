#!/usr/bin/env python3
"""Fixtures from the UNMODIFIED reference whose geometry enters the fused MFMA kernel
(2-D, C = 1, stride 1, P <= 7, M in {32, 64}; cdl_fused2d_supported) -- the fixtures of
make_golden.py all have M <= 8 and therefore only reach the shape-generic kernels.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_fused.py

Same import shim as make_golden.py (empty torchvision stub).  Seeds are screened for conditioning: a
pre-shrinkage value within rounding of its threshold makes the support of a code (and with it single
gradient entries) a coin flip between two correct evaluations, so the reference is run in fp32 AND in
fp64 (`net.double()`, same unmodified classes) and the first seed whose fp32 gradients agree with the fp64
ones to 2e-6 of each tensor's maximum is kept (`grad_cond` in the fixture; the stored outputs are the
fp32 ones).
"""
import copy

import numpy as np
import torch

from make_golden import grads_of, import_reference, perturb, save, smooth, state


def make(CDLNet, K, M, P, shape, per_sample, seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1000)
    net = CDLNet(K=K, M=M, P=P, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    perturb(net, t_lo=1e-2, t_hi=5e-2)
    x = smooth(shape, g)
    sig = (torch.tensor([18.0, 31.0]).reshape(2, 1, 1, 1) if per_sample else 25.0)
    y = x + torch.randn(x.shape, generator=g) * sig / 255
    return net, x, y, sig


def conditioning(net, x, y, sig):
    """max over parameters of |grad_fp32 - grad_fp64| / max|grad_fp64| for the reference itself."""
    n32, n64 = copy.deepcopy(net), copy.deepcopy(net).double()
    xh, _ = n32(y, sig)
    torch.mean((x - xh) ** 2).backward()
    xh64, _ = n64(y.double(), sig.double() if torch.is_tensor(sig) else sig)
    torch.mean((x.double() - xh64) ** 2).backward()
    worst = 0.0
    for (_, p), (_, q) in zip(n32.named_parameters(), n64.named_parameters()):
        if p.grad is not None:
            worst = max(worst, float((p.grad.double() - q.grad).abs().max() / q.grad.abs().max()))
    return worst


def main():
    CDLNet = import_reference()[0].CDLNet

    # ---- F10: K=3 M=32 P=7, 2 x 1 x 40 x 72 (ragged 64 x 32 tiles both ways), per-sample sigma
    # ---- F11: K=2 M=64 P=5, 1 x 1 x 32 x 64 (exactly one tile), float sigma
    for name, K, M, P, shape, per_sample in (("f10_fused_m32_p7", 3, 32, 7, (2, 1, 40, 72), True),
                                             ("f11_fused_m64_p5", 2, 64, 5, (1, 1, 32, 64), False)):
        for seed in range(100, 140):
            net, x, y, sig = make(CDLNet, K, M, P, shape, per_sample, seed)
            m = conditioning(net, x, y, sig)
            if m < 2e-6:
                break
        else:
            raise SystemExit(f"{name}: no well-conditioned seed")
        codes = list(net.forward_generator(y, sig))
        xhat, z = net(y, sig)
        loss = torch.mean((x - xhat) ** 2)
        loss.backward()
        print(f"{name}: seed {seed}, fp32-vs-fp64 gradient agreement {m:.3e}, nnz {float((z != 0).float().mean()):.3f}")
        save(name, x=x, y=y, sigma=sig, xhat=xhat, z=z, loss=loss, grad_cond=m, seed=seed,
             code0=codes[0], **state(net), **grads_of(net),
             hyper=np.array([K, M, P, 1, 1]), t0=5e-3)


if __name__ == "__main__":
    main()
