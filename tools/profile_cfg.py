#!/usr/bin/env python3
"""One forward+backward of a BASELINE configuration for rocprofv3 (tools/bench_configs.py CONFIGS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402
from bench_configs import CONFIGS               # noqa: E402

name = sys.argv[1]
kind, kw, shape, sigma, masked, _ = CONFIGS[name]
torch.manual_seed(1)
cls = {"2d": cva.CDLNet, "3d": cva.CDLNetVideo, "gabor": cva.GDLNet}[kind]
extra = {"depth": shape[2]} if kind == "3d" else {}
net = cls(**kw, t0=5e-3, adaptive=True, init=False, **extra)
with torch.no_grad():
    for p in net.parameters():
        if p.dim() > 3:
            p.mul_(0.02)
net = net.cuda()
x = torch.rand(shape, device="cuda")
mask = cva.gen_bayer_mask(x.cpu()).cuda() if masked else 1
y = mask * (x + torch.randn_like(x) * 25 / 255)
for _ in range(2):
    for p in net.parameters():
        p.grad = None
    xhat, _ = net(y, 25.0, mask=mask)
    torch.mean((x - xhat) ** 2).backward()
torch.cuda.synchronize()
