#!/usr/bin/env python3
"""Is k_stage<FWD> bound by the memory system or by the CU?  Time one launch with the persistent grid
limited to G workgroups (CDL_FUSED_GRID) on N images: N=64 streams 2.1 GB from HBM, N=8 (268 MB of fat
traffic, re-run back to back) stays in the 256 MiB Infinity Cache.  Per-tile time per workgroup =
launch time / (tiles / G)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from cdlnet_video_amd import _lib, ops          # noqa: E402

M, P, H, W = 64, 7, 256, 256
gen = torch.Generator(device="cuda").manual_seed(3)
for N in (64, 8, 4):
    g = ops.Geometry.make(N, 1, M, (H, W), (P, P), (3, 3), 1)
    r = torch.randn(N, 1, H, W, device="cuda", generator=gen)
    z = torch.randn(N, M, H, W, device="cuda", generator=gen) * (torch.rand(N, M, H, W, device="cuda", generator=gen) < 0.3)
    tau = torch.full((N, M), 0.5, device="cuda")
    wA = 0.05 * torch.randn(M, 1, P, P, device="cuda", generator=gen)
    frags = ops.fused_prep(wA, wA)
    patches = ops.fused_patches(g, r.device)
    out = torch.empty_like(z)
    tiles = N * 4 * 8
    for G in (256, 128, 64, 32):
        if G > tiles:
            continue
        os.environ["CDL_FUSED_GRID"] = str(G)
        _lib.reload_options()
        for _ in range(3):
            ops.fused_iter(g, r, z, tau, frags, -1.0, patches, "split3", out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            ops.fused_iter(g, r, z, tau, frags, -1.0, patches, "split3", out=out)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        per_tile_us = ms * 1e3 / (tiles / G)
        gbs = 2 * N * M * H * W * 4 / ms / 1e6
        print(json.dumps({"N": N, "grid": G, "tiles_per_wg": tiles / G, "ms": round(ms, 4),
                          "us_per_tile_per_wg": round(per_tile_us, 2), "GBps": round(gbs, 1),
                          "GBps_per_wg": round(gbs / G, 2)}), flush=True)
