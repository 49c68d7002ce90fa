#!/usr/bin/env python3
"""GPU-box diagnostic: per-parameter gradient error of the cfg4 (C=3 + Bayer mask, K=42) net against the oracle on
identical support, with the matrix-core kernels of the generic tier on and off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cdlnet_video_amd as cva
from oracle import cdl_oracle as O

K = int(sys.argv[1]) if len(sys.argv) > 1 else 42
torch.manual_seed(4)
M, P = 64, 7
net = cva.CDLNet(K=K, M=M, P=P, s=1, C=3, t0=5e-3, adaptive=True, init=True)
with torch.no_grad():
    for n_, p in net.named_parameters():
        if n_ not in ("t", "g"):
            p.add_(0.03 * p.abs().mean() * torch.randn_like(p))
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.cuda()
shape = (1, 3, 128, 128)
x = cva.utils.synthetic_clip(shape, seed=4)
m = cva.gen_bayer_mask(x)
sig = torch.tensor([12.0]).reshape(1, 1, 1, 1)
y = m * (x + torch.randn(shape, generator=torch.Generator().manual_seed(104)) * sig / 255)
ref64 = None
try:                                            # how far the fp32 oracle itself is from an fp64 evaluation
    sd64 = {k: v.double() for k, v in sd.items()}
    net._run  # noqa
except Exception:
    pass
for envs in ({}, {"CDL_FUSEDG_PREC": "0"}, {"CDL_FUSEDG_PREC": "2", "CDL_MFMA_WGRAD": "0"}, {"CDL_FUSEDG_PREC": "0", "CDL_MFMA_WGRAD": "0"}):
    for k in ("CDL_MFMA_SYNTHESIS", "CDL_MFMA_WGRAD", "CDL_MFMA_ANALYSIS", "CDL_FUSEDG_PREC"):
        os.environ.pop(k, None)
    os.environ.update(envs)
    cva._lib.reload_options()
    for p in net.parameters():
        p.grad = None
    outs = net._run(y.cuda(), sig.cuda(), m.cuda(), True)
    xhat, zK = outs[0], outs[1]
    codes = [c.detach().cpu() for c in outs[2:]] + [zK.detach().cpu()]
    torch.mean((x.cuda() - xhat) ** 2).backward()
    lref, grads, xref = O.loss_and_grads(sd, x, y, K=K, P=P, s=1, sigma=sig, adaptive=True, mask=m, supports=codes)
    errs = {n: float((p.grad.cpu() - grads[n]).abs().max() / grads[n].abs().max()) for n, p in net.named_parameters() if n != "g"}
    mags = {n: float(grads[n].abs().max()) for n in errs}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(envs, "worst:", [(n, f"{e:.2e}", f"max|g|={mags[n]:.1e}") for n, e in worst], flush=True)
    print("   B.1..B.4:", [f"{errs[f'B.{k}.weight']:.2e}" for k in range(1, min(K, 5))], "A.1:", f"{errs['A.1.weight']:.2e}", flush=True)
