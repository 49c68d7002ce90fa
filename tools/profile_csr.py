#!/usr/bin/env python3
"""Small CSR inference workload for rocprofv3 (argscsr.json net, one 256x256 frame, both-neighbour call)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

torch.manual_seed(1)
net = cva.CDLNet_CSRf2(K=30, M=169, P=9, s=2, C=1, t0=5e-3, adaptive=True, init=False).cuda()
with torch.no_grad():
    for k in range(30):
        net.A[k].weight.mul_(0.02)
        net.B[k].weight.mul_(0.02)
    net.g1.fill_(0.5)
    net.g2.fill_(0.5)
y = torch.rand(1, 1, 256, 256, device="cuda")
with torch.no_grad():
    _, z0 = net(y, None, None, 25.0)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        net(y, z0, z0, 25.0)
torch.cuda.synchronize()
