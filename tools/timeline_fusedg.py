#!/usr/bin/env python3
"""GPU-box: cycle-stamped timeline of workgroup 0 of one fused generic stage launch (cdl_fusedg_set_timeline):
where a tile's time goes, per wave.  Stamps (s_memtime, shader clocks), per tile:
  top | barrier 1 | staged (thin planes + patch zeroing in LDS) | barrier 2 |
  per row block: analysis done | epilogue done | synthesis + col2im done |
  ring flushed | next tile's thin loads issued | barrier 3 | patches combined

    python tools/timeline_fusedg.py cfg3|cfg4
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(3):
    o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out)
torch.cuda.synchronize()
tl = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
lib = cva._lib.lib()
lib.cdl_fusedg_set_timeline(tl.data_ptr())
o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out)
torch.cuda.synchronize()
lib.cdl_fusedg_set_timeline(None)
t = tl.view(8, 256).cpu().numpy()
t0 = t[:, 0].min()
names = ["top", "barrier1", "staged", "barrier2"]
for b in range(4):
    names += [f"b{b}.analysis", f"b{b}.epilogue", f"b{b}.synthesis"]
names += ["ring", "next loads issued", "barrier3", "combined"]
per_tile = len(names)
ntiles = 0
while 1 + (ntiles + 1) * per_tile <= 256 and t[0, 1 + (ntiles + 1) * per_tile - 1] > 0:
    ntiles += 1
print(json.dumps({"shape": cfg, "tiles_of_workgroup_0": ntiles, "total_cycles": int(t[:, :1 + ntiles * per_tile].max() - t0)}))
for tile in range(ntiles):
    base = 1 + tile * per_tile
    prev = t[:, base - 1]
    for i, nm in enumerate(names):
        cur = t[:, base + i]
        d = cur - prev
        print(json.dumps({"tile": tile, "phase": nm, "mean_cycles": int(d.mean()), "min": int(d.min()), "max": int(d.max()),
                          "end_mean": int((cur - t0).mean())}))
        prev = cur
