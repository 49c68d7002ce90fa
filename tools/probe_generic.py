#!/usr/bin/env python3
"""GPU-box: the generic-tier matrix-core analysis (+ST epilogue) and synthesis launches alone at the shipped s2030
shape (M = 169, P = 7, stride 2, 64 x 256 x 256), for rocprofv3.

    python tools/probe_generic.py [launches]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, C, M, sp, P, s = 64, 1, 169, (256, 256), (7, 7), 2
g = o.Geometry.make(N, C, M, sp, P, (3, 3), s)
gen = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(g.image_shape(), device="cuda", generator=gen)
z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
tau = torch.full((N, M), 0.3, device="cuda")
out = torch.empty_like(z)
thin = torch.empty_like(r)
fat = z.numel() * 4


def ev(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = ev(lambda: o.analysis(g, r, w, -1.0, zin=z, tau=tau, out=out))
print(json.dumps({"op": "analysis + ST (z' = ST(z - A r))", "ms": round(ms, 4), "GBps": round(2 * fat / ms / 1e6, 1)}))
ms = ev(lambda: o.synthesis(g, z, w, 1.0, out=thin))
print(json.dumps({"op": "synthesis (B z)", "ms": round(ms, 4), "GBps": round(fat / ms / 1e6, 1)}))
