#!/usr/bin/env python3
"""GPU-box ablation of the strip kernel (cdl_strip.hip) at the shipped CDLNet-s2030 shape (64 x 256 x 256, M=169, P=7,
s=2): time one launch with parts of the kernel switched off at run time (CDL_FUSED_DEBUG bits of the PROBE build of the
library; results are wrong, only the time matters).

    python tools/probe_strip.py [fwd|bwd] [nchw|rsc]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ablate                                  # noqa: E402,F401  (probe build of the library)
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
lay = sys.argv[2] if len(sys.argv) > 2 else "rsc"         # layout of the fat operands (rsc: what the sweeps use)
N, M, P, s, sp = 64, 169, 7, 2, (256, 256)
g = o.Geometry.make(N, 1, M, sp, (P, P), (3, 3), s)
gen = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(g.image_shape(), device="cuda", generator=gen)
z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
tau = torch.full((N, M), 0.3, device="cuda")
frags = o.fusedg_prep(g, w, w)
patches = o.fusedg_patches(g, "cuda")
out = torch.empty_like(z)
bits = o.fusedg_support_map(g, z)            # (from the reference-layout tensor, before any conversion)
dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
fat = z.numel() * 4
names = {0: "full kernel", 1: "no fat loads", 2: "no fat stores", 3: "no fat traffic", 4: "no analysis-like MFMAs",
         8: "no synthesis-like MFMAs", 12: "no MFMAs at all", 16: "no col2im", 32: "no im2col gather",
         60: "no MFMA, col2im, gather (epilogue + traffic)", 63: "row loop skeleton + epilogue VALU only"}
if lay == "rsc":
    z, out = o.fusedg_to_rsc(g, z), o.fusedg_rsc_buffer(g, "cuda")[0]
if mode == "fwd":
    fn = lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out, lay_in=lay, lay_out=lay)
else:
    fn = lambda: o.fusedg_stage_bwd(g, r, z, bits, frags, patches, dtp, True, out=out, lay_in=lay, lay_out=lay)
rows = []
for rnd in range(3):
    for dbg, name in names.items():
        os.environ["CDL_FUSED_DEBUG"] = str(dbg)
        cva._lib.reload_options()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        rows.append((dbg, name, a.elapsed_time(b) / 10))
os.environ["CDL_FUSED_DEBUG"] = "0"
cva._lib.reload_options()
for dbg, name in names.items():
    ms = sorted(t for d, _, t in rows if d == dbg)[1]
    print(json.dumps({"shape": "s2030 64x256x256", "mode": mode, "layout": lay, "debug_bits": dbg, "variant": name, "ms": round(ms, 4),
                      "GBps_if_full_traffic": round(2 * fat / ms / 1e6, 1)}), flush=True)
