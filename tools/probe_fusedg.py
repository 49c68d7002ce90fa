#!/usr/bin/env python3
"""GPU-box ablation of the fused generic stage (cdl_fusedg.hip) at a BASELINE shape: time one forward launch with
parts of the kernel switched off at run time (CDL_FUSED_DEBUG bits; results are wrong, only the time matters).

    python tools/probe_fusedg.py cfg3|cfg4
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'probes' else '.'))
import _ablate                                  # noqa: E402,F401  (probe build of the library)
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if cfg == "cfg3":
    N, C, M, sp, P = 8, 1, 48, (8, 128, 128), (5, 5, 5)
else:
    N, C, M, sp, P = 8, 3, 64, (256, 256), (7, 7)
g = o.Geometry.make(N, C, M, sp, P, tuple(p // 2 for p in P), 1)
gen = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(g.image_shape(), device="cuda", generator=gen)
z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
tau = torch.full((N, M), 0.3, device="cuda")
frags = o.fusedg_prep(g, w, w)
patches = o.fusedg_patches(g, "cuda")
out = torch.empty_like(z)
fat = z.numel() * 4
tile = os.environ.get("CDL_FUSEDG_STRIP", "0") in ("", "0")
if tile:      # the tile kernel k_stage_g (cdl_fusedg.hip)
    names = {0: "full kernel", 1: "no thin staging after the first tile", 4: "no synthesis / col2im", 8: "no fat loads",
             16: "no fat stores", 24: "no fat traffic", 32: "no patch combine", 61: "tile loop skeleton only"}
else:         # the strip kernel k_stripg (cdl_stripg.hip), the default
    names = {0: "full kernel", 1: "no fat loads", 2: "no fat stores", 3: "no fat traffic", 4: "no analysis-like MFMAs",
             8: "no synthesis-like MFMAs", 12: "no MFMAs at all", 16: "no col2im", 32: "no im2col gather",
             60: "no MFMA, col2im, gather (epilogue + traffic)", 63: "row loop skeleton + epilogue VALU only"}
rows = []
for rnd in range(3):
    for dbg, name in names.items():
        os.environ["CDL_FUSED_DEBUG"] = str(dbg)
        cva._lib.reload_options()
        for _ in range(2):
            o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out)
        b.record()
        torch.cuda.synchronize()
        rows.append((dbg, name, a.elapsed_time(b) / 10))
os.environ["CDL_FUSED_DEBUG"] = "0"
cva._lib.reload_options()
for dbg, name in names.items():
    ms = sorted(t for d, _, t in rows if d == dbg)[1]
    print(json.dumps({"shape": cfg, "kernel": "k_stage_g (tile)" if tile else "k_stripg (strip)", "debug_bits": dbg, "variant": name, "ms": round(ms, 4),
                      "GBps_if_full_traffic": round(2 * fat / ms / 1e6, 1)}), flush=True)
