#!/usr/bin/env python3
"""GPU-box workload for tools/profile_workload.sh: the launches of one forward / reverse iteration of the shipped
CDLNet-s2030 architecture (K=30 M=169 P=7 s=2) at 64 x 256 x 256, internal codes in CDL_LAY_RSC -- the strip kernel's
forward stage (with the training map), its reverse stage, the paired filter-gradient launch and the thin assemble.

    python tools/probe_s2030.py [reps]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, M, P, s, sp = 64, 169, 7, 2, (256, 256)
g = o.Geometry.make(N, 1, M, sp, (P, P), (3, 3), s)
gen = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(g.image_shape(), device="cuda", generator=gen)
z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
tau = torch.full((N, M), 0.3, device="cuda")
assert o.fusedg_code_layout(g) == "rsc"
frags = o.fusedg_prep(g, w, w)
patches = o.fusedg_patches(g, "cuda")
bits = o.fusedg_support_map(g, z)
zr, dur, outr = o.fusedg_to_rsc(g, z), o.fusedg_to_rsc(g, z * 0.5), o.fusedg_rsc_buffer(g, "cuda")[0]
dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
thin = torch.empty_like(r)
# the paired filter gradient on rsc operands goes through the reverse sweep's C entry point only: one K = 2 sweep per rep
fat = z.numel() * 4
print(f"algorithmic bytes per launch: stage FWD {2 * fat + bits.numel() * 4 + 2 * r.numel() * 4}, "
      f"stage BWD {2 * fat + bits.numel() * 4 + 2 * r.numel() * 4}, filter-gradient pair {2 * fat + 2 * r.numel() * 4}", flush=True)
for _ in range(reps):
    o.fusedg_iter(g, r, zr, tau, frags, -1.0, patches, out=outr, map_out=bits, lay_in="rsc", lay_out="rsc")
    o.fusedg_assemble(g, patches, None, r, 1.0, out=thin)
    o.fusedg_stage_bwd(g, r, dur, bits, frags, patches, dtp, True, out=outr, lay_in="rsc", lay_out="rsc")
    o.fusedg_assemble(g, patches, None, None, -1.0, out=thin)
torch.cuda.synchronize()
# one short training step of the net itself: the paired filter-gradient launches on rsc operands
torch.manual_seed(1)
net = cva.CDLNet(K=3, M=M, P=P, s=s, C=1, t0=5e-3, adaptive=True, init=False).cuda()
x = torch.rand(g.image_shape(), device="cuda")
for _ in range(reps):
    for p_ in net.parameters():
        p_.grad = None
    xhat, _ = net(x, 25.0)
    torch.mean((x - xhat) ** 2).backward()
torch.cuda.synchronize()
print("done", flush=True)
