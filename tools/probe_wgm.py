#!/usr/bin/env python3
"""GPU-box: the matrix-core filter-gradient launch (cdl_wgrad_mfma.hip) alone at a BASELINE shape, for rocprofv3.

    python tools/probe_wgm.py cfg3|cfg4 [launches]
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if cfg == "cfg3":
    N, C, M, sp, P = 8, 1, 48, (8, 128, 128), (5, 5, 5)
else:
    N, C, M, sp, P = 8, 3, 64, (256, 256), (7, 7)
g = o.Geometry.make(N, C, M, sp, P, tuple(p // 2 for p in P), 1)
gen = torch.Generator(device="cuda").manual_seed(0)
r = torch.randn(g.image_shape(), device="cuda", generator=gen)
z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
fat = z.numel() * 4
names = {0: "full", 1024: "no thin staging", 2048: "no k-loop", 4096: "no epilogue", 1024 + 4096: "k-loop only",
         2048 + 4096: "staging only", 1024 + 2048: "epilogue only", 7168: "skeleton"}
variants = names if os.environ.get("WGM_ABLATE") else {0: "full"}
for dbg, name in variants.items():
    os.environ["CDL_FUSED_DEBUG"] = str(dbg | int(os.environ.get("WGM_BASE", "0")))
    cva._lib.reload_options()
    for _ in range(3):
        dw = o.wgrad(g, z, r, -1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dw = o.wgrad(g, z, r, -1.0)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(json.dumps({"shape": cfg, "variant": name, "wgrad_ms": round(ms, 4), "GBps": round(fat / ms / 1e6, 1),
                      "frac_of_8TBps": round(fat / ms / 8e9, 3)}), flush=True)
os.environ["CDL_FUSED_DEBUG"] = "0"
cva._lib.reload_options()
