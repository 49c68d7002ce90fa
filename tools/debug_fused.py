#!/usr/bin/env python3
"""Where does the fused iteration disagree with the generic kernels? (GPU box diagnostic.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                        # noqa: E402
import cdlnet_video_amd as cva      # noqa: E402


def run(N, M, P, H, W, seed):
    o = cva.ops
    gen = torch.Generator().manual_seed(seed)
    geom = o.Geometry.make(N, 1, M, (H, W), (P, P), (P // 2, P // 2), 1)
    r = torch.randn(N, 1, H, W, generator=gen).cuda()
    z = (torch.randn(N, M, H, W, generator=gen) * (torch.rand(N, M, H, W, generator=gen) < 0.3)).cuda()
    wA = (torch.randn(M, 1, P, P, generator=gen) * 0.15).cuda()
    wB = (torch.randn(M, 1, P, P, generator=gen) * 0.15).cuda()
    tau = (torch.rand(N, M, generator=gen) * 0.6 - 0.05).cuda()
    frags = o.fused_prep(wA, wB)
    patches = o.fused_patches(geom, "cuda")
    for name, zin, sgn in (("iter", z, -1.0), ("first", None, 1.0)):
        ref = o.analysis(geom, r, wA, sgn, zin, None, tau)
        pre = o.analysis(geom, r, wA, sgn, zin, None, None)            # u before shrinkage
        got = o.fused_iter(geom, r, zin, tau, frags, sgn, patches, "split3")
        d = (got - ref).abs()
        bad = (d > 1e-4 * ref.abs().max()).nonzero()
        print(f"[{N},{M},{P},{H}x{W}] {name}: max err {float(d.max()):.3e} (ref max {float(ref.abs().max()):.2f}) "
              f"bad={len(bad)}")
        for idx in bad[:12].tolist():
            n, ch, y, x = idx
            print(f"   n={n} ch={ch} y={y} x={x} got={float(got[n,ch,y,x]):+.6f} ref={float(ref[n,ch,y,x]):+.6f} "
                  f"u={float(pre[n,ch,y,x]):+.6f} tau={float(tau[n,ch]):+.4f} zin={0.0 if zin is None else float(zin[n,ch,y,x]):+.5f}")
        if len(bad):
            b = bad.float()
            print("   y range", int(b[:, 2].min()), int(b[:, 2].max()), "x range", int(b[:, 3].min()), int(b[:, 3].max()),
                  "n", sorted(set(bad[:, 0].tolist())), "ch count", len(set(bad[:, 1].tolist())))


if __name__ == "__main__":
    import collections
    o = cva.ops
    N, M, P, H, W = 1, 64, 7, 128, 128
    gen = torch.Generator().manual_seed(5)
    geom = o.Geometry.make(N, 1, M, (H, W), (P, P), (3, 3), 1)
    r = torch.randn(N, 1, H, W, generator=gen).cuda()
    z = torch.randn(N, M, H, W, generator=gen).cuda()
    wA = (torch.randn(M, 1, P, P, generator=gen) * 0.15).cuda()
    tau = torch.full((N, M), 0.3).cuda()
    frags = o.fused_prep(wA, wA)
    patches = o.fused_patches(geom, "cuda")
    ref = o.analysis(geom, r, wA, -1.0, z, None, tau)
    got = o.fused_iter(geom, r, z, tau, frags, -1.0, patches, "split3")
    d = (got - ref).abs()
    bad = (d > 1e-3).nonzero()
    print("bad", len(bad), "of", d.numel())
    if len(bad):
        ys = collections.Counter((bad[:, 2] // 8).tolist()); xs = collections.Counter((bad[:, 3] // 4).tolist())
        chs = collections.Counter(bad[:, 1].tolist())
        print("by row block (y//8):", sorted(ys.items()))
        print("by x quad (x//4) first 12:", sorted(xs.items())[:12])
        print("by channel first 12:", sorted(chs.items())[:12])
        for idx in bad[:6].tolist():
            n, ch, y, x = idx
            print(f"  ch={ch} y={y} x={x} got={float(got[n,ch,y,x]):+.4f} ref={float(ref[n,ch,y,x]):+.4f}")
