#!/usr/bin/env python3
"""Kernel-level timing on the GPU box: each fat kernel of the fused path at the cfg2 shape, HIP-event timed, in
every code layout (reference NCHW, pixel-blocked fp32, pixel-blocked bf16 storage), the variants interleaved
in rounds inside ONE process (median over rounds).  One JSON object per line on stdout.

    BK_LAYOUTS=nchw,blocked,blocked_bf16 BK_ROUNDS=5 python tools/bench_kernels.py
    BK_LAYOUTS=blocked BK_ROUNDS=1 ...        (what the rocprofv3 --pmc passes run: one layout, few launches)
"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                        # noqa: E402
import cdlnet_video_amd as cva      # noqa: E402


def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N = int(os.environ.get("BK_N", 64)); S = int(os.environ.get("BK_S", 256)); M, P = 64, 7
    layouts = os.environ.get("BK_LAYOUTS", "nchw,blocked,blocked_bf16").split(",")
    rounds = int(os.environ.get("BK_ROUNDS", 5)); reps = int(os.environ.get("BK_REPS", 5))
    o = cva.ops
    g = o.Geometry.make(N, 1, M, (S, S), (P, P), (3, 3), 1)
    dev = "cuda"
    r = torch.randn(g.image_shape(), device=dev)
    z = torch.randn(g.code_shape(), device=dev) * (torch.rand(g.code_shape(), device=dev) < 0.2)
    gup = torch.randn(g.code_shape(), device=dev)
    tau = torch.full((N, M), 0.3, device=dev)
    w = torch.randn(M, 1, P, P, device=dev) * 0.1
    thin = torch.empty_like(r)
    th = r.numel() * 4
    frags = o.fused_prep(w, w)
    patches = o.fused_patches(g, dev)
    bits = o.fused_support_map(g, z)
    mapb = bits.numel() * 4
    dtp = torch.empty((o.fused_tiles(g), M), device=dev)
    ws = o.fused_wgrad_workspace(g, dev)
    cases = {}
    for lay in layouts:
        zl, gl = o.fused_from_nchw(g, z, lay), o.fused_from_nchw(g, gup, lay)
        out = o.fused_code_buffer(g, lay, dev)[0]
        fat = out.numel() * out.element_size()
        kw = dict(lay_in=lay, lay_out=lay)
        for prec in ("split3", "bf16"):
            cases[f"k_stage<FWD>[{prec},{lay},+map]"] = (
                lambda zl=zl, out=out, prec=prec, kw=kw: o.fused_iter(g, r, zl, tau, frags, -1.0, patches, prec, out=out, map_out=bits, **kw),
                2 * fat + mapb + 2 * th)
            # the forms cdl_fused2d_backward launches: dA_k riding in the reverse stage, dB_k alone in k_wgrad2d
            cases[f"k_stage<BWD>[{prec},{lay},+dA]"] = (
                lambda gl=gl, out=out, prec=prec, kw=kw: o.fused_stage_bwd(g, r, gl, bits, frags, patches, dtp, True, prec, out=out,
                                                                           r2=r, alpha=-1.0, workspace=ws, **kw),
                2 * fat + mapb + 3 * th)
            cases[f"k_wgrad2d[{prec},{lay},dB only]"] = (
                lambda zl=zl, prec=prec, lay=lay: o.fused_wgrad(g, ws, zl, r, 1.0, precision=prec, layout=lay),
                fat + th)
        cases[f"k_stage<FWD>[split3,{lay},no map]"] = (
            lambda zl=zl, out=out, kw=kw: o.fused_iter(g, r, zl, tau, frags, -1.0, patches, "split3", out=out, **kw), 2 * fat + 2 * th)
        cases[f"k_stage<FIRST>[split3,{lay}]"] = (
            lambda out=out, lay=lay: o.fused_iter(g, r, None, tau, frags, 1.0, patches, "split3", out=out, lay_out=lay), fat + 2 * th)
    outn = torch.empty_like(z)
    cases["torch copy fat (yardstick)"] = (lambda: outn.copy_(z), 2 * z.numel() * 4)
    cases["k_assemble"] = (lambda: o.fused_assemble(g, patches, None, r, 1.0, out=thin), 3 * th)
    dt = torch.zeros(2, M, device=dev)
    cases["k_dtau_reduce"] = (lambda: o.fused_dtau_reduce(g, dtp, None, dt), 0)
    cases["k_prep"] = (lambda: o.fused_prep(w, w), 0)
    for fn, _ in cases.values():            # warm-up (LDS attributes, allocator)
        fn()
    torch.cuda.synchronize()
    times = {k: [] for k in cases}
    for _ in range(rounds):
        for k, (fn, _) in cases.items():
            times[k].append(ev(fn, reps))
    for k, (fn, nbytes) in cases.items():
        ms = statistics.median(times[k])
        row = {"kernel": k, "ms": round(ms, 4), "min_ms": round(min(times[k]), 4), "rounds": rounds}
        if nbytes:
            row.update(alg_bytes=nbytes, GBps=round(nbytes / ms / 1e6, 1), frac_of_8TBps=round(nbytes / ms / 8e9, 3))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
