#!/usr/bin/env python3
"""GPU-box scratch diagnostics for the strip kernel (cdl_strip.hip): mismatch positions of the reverse stage and
per-launch times at the s2030 benchmark shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops


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


def bwd_mismatch():
    N, M, P, s, sp = 2, 169, 7, 2, (64, 128)
    gen = torch.Generator().manual_seed(7 * sum(sp) + M)
    g = o.Geometry.make(N, 1, M, sp, (P, P), (3, 3), s)
    ish, csh, fsh = g.image_shape(), g.code_shape(), g.filter_shape()
    thin = torch.randn(ish, generator=gen).cuda()
    base = torch.randn(csh, generator=gen).cuda()
    gate = (torch.randn(csh, generator=gen) * (torch.rand(csh, generator=gen) < 0.3)).cuda()
    w1 = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    w2 = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    frags = o.fusedg_prep(g, w1, w2)
    patches = o.fusedg_patches(g, "cuda")
    dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
    bits = o.fusedg_support_map(g, gate)
    gk = o.analysis(g, thin, w1, 1.0, base, None, None)
    du_ref = gk * (gate != 0)
    out = torch.full(csh, 777.0, device="cuda")
    du = o.fusedg_stage_bwd(g, thin, base, bits, frags, patches, dtp, True, out=out)
    bad = (du == 0) != (du_ref == 0)
    print("never written:", int((du == 777.0).sum()), "zero-pattern mismatches:", int(bad.sum()), "of", bad.numel())
    idx = bad.nonzero()[:20]
    for i in idx.tolist():
        n, ch, y, x = i
        print(i, "du", float(du[n, ch, y, x]), "ref", float(du_ref[n, ch, y, x]), "gate", float(gate[n, ch, y, x]), "gk", float(gk[n, ch, y, x]))
    if len(idx):
        print("channels:", sorted(set(bad.nonzero()[:, 1].tolist()))[:40])
        print("rows:", sorted(set(bad.nonzero()[:, 2].tolist()))[:40])
        print("cols:", sorted(set(bad.nonzero()[:, 3].tolist()))[:70])


def launch_times():
    N, M, P, s, sp = 64, 169, 7, 2, (256, 256)
    g = o.Geometry.make(N, 1, M, sp, (P, P), (3, 3), s)
    gen = torch.Generator(device="cuda").manual_seed(0)
    r = torch.randn(g.image_shape(), device="cuda", generator=gen)
    z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.2)
    w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.05
    tau = torch.full((N, M), 0.3, device="cuda")
    print("fusedg_supported", o.fusedg_supported(g), "sizes", o._fusedg_sizes(g))
    frags = o.fusedg_prep(g, w, w)
    patches = o.fusedg_patches(g, "cuda")
    out = torch.empty_like(z)
    bits = o.fusedg_map(g, "cuda")
    dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
    thin = torch.empty_like(r)
    fat = z.numel() * 4
    zr, outr = o.fusedg_to_rsc(g, z), o.fusedg_rsc_buffer(g, "cuda")[0]
    chk = o.fusedg_from_rsc(g, o.fusedg_iter(g, r, zr, tau, frags, -1.0, patches, out=outr, lay_in="rsc", lay_out="rsc"))
    ref = o.fusedg_iter(g, r, z, tau, frags, -1.0, patches)
    print("rsc == nchw bit for bit:", bool(torch.equal(chk, ref)), flush=True)
    for name, fn, nbytes in (
            ("strip FWD rsc->rsc", lambda: o.fusedg_iter(g, r, zr, tau, frags, -1.0, patches, out=outr, lay_in="rsc", lay_out="rsc"), 2 * fat),
            ("strip FWD rsc->rsc + map", lambda: o.fusedg_iter(g, r, zr, tau, frags, -1.0, patches, out=outr, map_out=bits, lay_in="rsc", lay_out="rsc"), 2 * fat + bits.numel() * 4),
            ("strip BWD rsc->rsc", lambda: o.fusedg_stage_bwd(g, r, zr, bits, frags, patches, dtp, True, out=outr, lay_in="rsc", lay_out="rsc"), 2 * fat + bits.numel() * 4),
            ("strip FWD nchw->rsc", lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=outr, lay_out="rsc"), 2 * fat),
            ("strip FWD (no map)", lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out), 2 * fat),
            ("strip FWD + map", lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out, map_out=bits), 2 * fat + bits.numel() * 4),
            ("strip FIRST", lambda: o.fusedg_iter(g, r, None, tau, frags, 1.0, patches, out=out), fat),
            ("strip BWD", lambda: o.fusedg_stage_bwd(g, r, z, bits, frags, patches, dtp, True, out=out), 2 * fat + bits.numel() * 4),
            ("strip assemble", lambda: o.fusedg_assemble(g, patches, None, r, 1.0, out=thin), 0),
            ("generic analysis", lambda: o.analysis(g, r, w, -1.0, z, None, tau, out=out), 2 * fat),
            ("generic synthesis", lambda: o.synthesis(g, z, w, 1.0, None, None, r, out=thin), fat)):
        ms = ev(fn)
        print(f"{name:24s} {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["bwd", "time"]
    if "bwd" in what:
        bwd_mismatch()
    if "time" in what:
        launch_times()


def map_mismatch():
    N, M, P, s, sp = 2, 169, 7, 2, (64, 128)
    gen = torch.Generator().manual_seed(sum(sp) + M)
    g = o.Geometry.make(N, 1, M, sp, (P, P), (3, 3), s)
    ish, csh, fsh = g.image_shape(), g.code_shape(), g.filter_shape()
    r = torch.randn(ish, generator=gen).cuda()
    z = (torch.randn(csh, generator=gen) * (torch.rand(csh, generator=gen) < 0.3)).cuda()
    wA = (torch.randn(fsh, generator=gen) * 0.1).cuda()
    tau = (torch.rand(N, M, generator=gen) * 0.6 + 0.01).cuda()
    frags = o.fusedg_prep(g, wA, wA)
    patches = o.fusedg_patches(g, "cuda")
    for name, zin, sgn in (("iter", z, -1.0), ("first", None, 1.0)):
        bits = o.fusedg_map(g, "cuda").fill_(-1)
        zg = o.fusedg_iter(g, r, zin, tau, frags, sgn, patches, map_out=bits)
        ref = o.fusedg_support_map(g, zg)
        bad = bits != ref
        print(name, "map words differing:", int(bad.sum()), "of", bad.numel(), "per plane:", bad.sum(dim=(0, 2, 3, 4)).tolist())
        if bad.any():
            i = bad.nonzero()[0].tolist()
            n, pl, _, y, x = i
            print(" first:", i, hex(int(bits[n, pl, 0, y, x]) & 0xffffffff), hex(int(ref[n, pl, 0, y, x]) & 0xffffffff))
            j, hh, kind = pl // 4, (pl // 2) & 1, pl & 1
            chs = [32 * (2 * j + (b // 16)) + 8 * ((b % 16) // 4) + 4 * hh + (b % 4) for b in range(32)]
            vals = [float(zg[n, ch, y, x]) if ch < M else None for ch in chs]
            print(" plane", pl, "pair", j, "h", hh, "kind", kind, "values:", vals)
        print(" negative zeros in z':", int(((zg == 0) & torch.signbit(zg)).sum()), "zeros:", int((zg == 0).sum()))


if __name__ == "__main__" and "map" in sys.argv[1:]:
    map_mismatch()


def stripg_times(cfg="cfg3"):
    import cdlnet_video_amd as cva2
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
    fat = z.numel() * 4
    for tile in ("0", "1"):
        os.environ["CDL_FUSEDG_STRIP"] = "1" if tile == "0" else "0"
        cva2._lib.reload_options()
        frags = o.fusedg_prep(g, w, w)
        patches = o.fusedg_patches(g, "cuda")
        out = torch.empty_like(z)
        bits = o.fusedg_support_map(g, z)
        dtp = torch.empty(o._fusedg_sizes(g)[2], M, device="cuda")
        thin = torch.empty_like(r)
        print(cfg, "kernel:", "tile" if tile == "1" else "strip", "sizes", o._fusedg_sizes(g), "layout", o.fusedg_code_layout(g))
        rows = [("FWD nchw", lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out), 2 * fat),
                ("FWD nchw + map", lambda: o.fusedg_iter(g, r, z, tau, frags, -1.0, patches, out=out, map_out=bits), 2 * fat),
                ("FIRST", lambda: o.fusedg_iter(g, r, None, tau, frags, 1.0, patches, out=out), fat),
                ("BWD nchw", lambda: o.fusedg_stage_bwd(g, r, z, bits, frags, patches, dtp, True, out=out), 2 * fat),
                ("assemble", lambda: o.fusedg_assemble(g, patches, None, r, 1.0, out=thin), 0)]
        if tile == "0":
            zr, outr = o.fusedg_to_rsc(g, z), o.fusedg_rsc_buffer(g, "cuda")[0]
            rows += [("FWD rsc", lambda: o.fusedg_iter(g, r, zr, tau, frags, -1.0, patches, out=outr, lay_in="rsc", lay_out="rsc"), 2 * fat),
                     ("BWD rsc", lambda: o.fusedg_stage_bwd(g, r, zr, bits, frags, patches, dtp, True, out=outr, lay_in="rsc", lay_out="rsc"), 2 * fat)]
        for name, fn, nbytes in rows:
            ms = ev(fn)
            print(f"  {name:16s} {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
    os.environ["CDL_FUSEDG_STRIP"] = "0"
    cva2._lib.reload_options()


if __name__ == "__main__" and "stripg" in sys.argv[1:]:
    stripg_times("cfg3")
    stripg_times("cfg4")
