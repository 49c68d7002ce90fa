#!/usr/bin/env python3
"""Kernel-level timing on the GPU box: each iteration kernel at the cfg2 shape, HIP-event timed.
Writes one JSON object per line to stdout (progress on stderr)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                        # noqa: E402
import cdlnet_video_amd as cva      # noqa: E402


def ev(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N = int(os.environ.get("BK_N", 64)); S = int(os.environ.get("BK_S", 256)); M, P = 64, 7
    o = cva.ops
    g = o.Geometry.make(N, 1, M, (S, S), (P, P), (3, 3), 1)
    dev = "cuda"
    r = torch.randn(g.image_shape(), device=dev)
    z = torch.randn(g.code_shape(), device=dev) * (torch.rand(g.code_shape(), device=dev) < 0.2)
    tau = torch.full((N, M), 0.3, device=dev)
    w = torch.randn(M, 1, P, P, device=dev) * 0.1
    out = torch.empty_like(z)
    thin = torch.empty_like(r)
    fat, th = z.numel() * 4, r.numel() * 4
    frags = o.fused_prep(w, w)
    patches = o.fused_patches(g, dev)
    rows = []
    bits = o.fused_support_map(g, z)
    mapb = bits.numel() * 4
    for prec in ("split3", "bf16"):
        ms = ev(lambda: o.fused_iter(g, r, z, tau, frags, -1.0, patches, prec, out=out))
        rows.append({"kernel": f"k_iter_fwd[{prec}]", "ms": ms, "alg_bytes": 2 * fat + 2 * th,
                     "GBps": (2 * fat + 2 * th) / ms / 1e6, "Mpix_iter_per_s": N * S * S / ms / 1e3})
    ms = ev(lambda: o.fused_iter(g, r, z, tau, frags, -1.0, patches, "split3", out=out, map_out=bits))
    rows.append({"kernel": "k_iter_fwd[split3,+map]", "ms": ms, "alg_bytes": 2 * fat + mapb + 2 * th,
                 "GBps": (2 * fat + mapb + 2 * th) / ms / 1e6})
    ms = ev(lambda: o.fused_iter(g, r, None, tau, frags, 1.0, patches, "split3", out=out))
    rows.append({"kernel": "k_iter_fwd[split3,first]", "ms": ms, "alg_bytes": fat + 2 * th,
                 "GBps": (fat + 2 * th) / ms / 1e6})
    gup = torch.randn_like(z)
    dtp = torch.empty((o.fused_tiles(g), M), device=dev)
    ws = o.fused_wgrad_workspace(g, dev)
    for prec in ("split3", "bf16"):
        ms = ev(lambda: o.fused_stage_bwd(g, r, gup, bits, frags, patches, dtp, True, prec, out=out))
        rows.append({"kernel": f"k_stage<BWD>[{prec}]", "ms": ms, "alg_bytes": 2 * fat + mapb + 2 * th,
                     "GBps": (2 * fat + mapb + 2 * th) / ms / 1e6})
        ms = ev(lambda: o.fused_wgrad(g, ws, gup, r, -1.0, z, r, 1.0, prec))
        rows.append({"kernel": f"k_wgrad2d[{prec}]", "ms": ms, "alg_bytes": 2 * fat + 2 * th,
                     "GBps": (2 * fat + 2 * th) / ms / 1e6})
    dt = torch.zeros(2, M, device=dev)
    ms = ev(lambda: o.fused_dtau_reduce(g, dtp, None, dt))
    rows.append({"kernel": "k_dtau_reduce", "ms": ms})
    ms = ev(lambda: o.fused_assemble(g, patches, None, r, 1.0, out=thin))
    rows.append({"kernel": "k_assemble", "ms": ms, "alg_bytes": 3 * th, "GBps": 3 * th / ms / 1e6})
    ms = ev(lambda: o.fused_prep(w, w))
    rows.append({"kernel": "k_prep", "ms": ms})
    ms = ev(lambda: out.copy_(z))
    rows.append({"kernel": "torch copy fat (HBM yardstick)", "ms": ms, "alg_bytes": 2 * fat,
                 "GBps": 2 * fat / ms / 1e6})
    if os.environ.get("BK_GENERIC", "0") == "1":
        ms = ev(lambda: o.analysis(g, r, w, -1.0, z, None, tau, out=out), 3)
        rows.append({"kernel": "generic k_analysis", "ms": ms, "GBps": (2 * fat + th) / ms / 1e6})
        ms = ev(lambda: o.synthesis(g, z, w, 1.0, None, None, r, out=thin), 3)
        rows.append({"kernel": "generic k_synthesis", "ms": ms, "GBps": (fat + 2 * th) / ms / 1e6})
        ms = ev(lambda: o.wgrad(g, gup, r, -1.0, gate=z), 2)
        rows.append({"kernel": "generic k_wgrad(gated)", "ms": ms})
        ms = ev(lambda: o.tau_grad(g, gup, z, None, dt), 3)
        rows.append({"kernel": "generic k_tau", "ms": ms, "GBps": 2 * fat / ms / 1e6})
    for row in rows:
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
