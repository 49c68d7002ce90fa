#!/usr/bin/env python3
"""GPU-box: seeded random geometries through the matrix-core tier (analysis, synthesis, filter gradients incl. the paired
launch, the fused generic stage where it applies) against the fp32 VALU tier of the same library (CDL_MFMA_*=0), which the
`-m gpu` suite pins to the oracle.  Catches indexing slips at ragged / odd sizes the fixed test shapes do not reach.

    python tools/fuzz_parity.py [cases] [seed]
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

o = cva.ops


def setenv(v):
    for k in ("CDL_MFMA_ANALYSIS", "CDL_MFMA_SYNTHESIS", "CDL_MFMA_WGRAD"):
        os.environ[k] = v
    cva._lib.reload_options()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def run(cases=40, seed=0, verbose=True):
    rng = random.Random(seed)
    worst = {}
    fails = []
    for case in range(cases):
        three = rng.random() < 0.35
        s = rng.choice([1, 1, 2])
        P2 = rng.choice([3, 5, 7, 7, 9] if not three else [3, 5, 5])
        Pd = rng.choice([1, 3, 5]) if three else 1
        C = rng.choice([1, 1, 3, 2])
        M = rng.choice([8, 16, 24, 32, 48, 64, 40, 169 if not three else 56, 96, 5])
        if three:
            D = rng.randint(2, 7) * s
            sp = (D, s * rng.randint(20, 48), s * rng.randint(33, 80))
            P = (Pd, P2, P2)
        else:
            sp = (s * rng.randint(24, 90), s * rng.randint(33, 150))
            P = (P2, P2)
        code_px = 1
        for d in sp:
            code_px *= d // s
        N = max(1, min(48, (300000 * 4) // max(code_px * max(M, 16) // 8, 1)))
        N = rng.randint(max(1, N // 2), max(1, N))
        pad = tuple(p // 2 for p in P)
        g = o.Geometry.make(N, C, M, sp, P, pad, s)
        gen = torch.Generator(device="cuda").manual_seed(case)
        x = torch.randn(g.image_shape(), device="cuda", generator=gen)
        z = torch.randn(g.code_shape(), device="cuda", generator=gen) * (torch.rand(g.code_shape(), device="cuda", generator=gen) < 0.3)
        u = torch.randn(g.code_shape(), device="cuda", generator=gen)
        w = torch.randn(g.filter_shape(), device="cuda", generator=gen) * 0.1
        tau = torch.rand(N, M, device="cuda", generator=gen) * 0.4 + 0.05
        mask = (torch.rand(g.image_shape(), device="cuda", generator=gen) < 0.5).float()
        res = {}
        for mode in ("0", "1"):
            setenv(mode)
            r = {}
            r["ana"] = o.analysis(g, x, w, -1.0, z, None, None)            # no threshold: ST would amplify sign flips at 0
            r["ana_first"] = o.analysis(g, x, w, 1.0)
            r["ana_gate"] = o.analysis(g, x, w, 1.0, u, z, None)
            r["syn"] = o.synthesis(g, z, w, -1.0, None, mask, x)
            r["syn_gate"] = o.synthesis(g, u, w, 1.0, z)
            r["wg"] = o.wgrad(g, z, x, 1.0)
            r["wg_gate"] = o.wgrad(g, u, x, -1.0, gate=z)
            r["pair0"], r["pair1"] = o.wgrad_pair(g, u * (z != 0), x, -1.0, z, x, 1.0)
            dt = torch.zeros(2, M, device="cuda")
            r["rev"] = o.analysis_rev(g, x, w, 0.7, u, z, tau[:, 0].contiguous(), dt)
            r["rev_dt"] = dt
            res[mode] = r
        setenv("1")
        tag = f"N{N} C{C} M{M} {sp} P{P} s{s}"
        for k in res["0"]:
            e = rel(res["1"][k], res["0"][k])
            worst[k] = max(worst.get(k, 0.0), e)
            if not e < 3e-5:
                fails.append((tag, k, e))
        # fused generic stage against the generic sweep kernels
        if o.fusedg_supported(g):
            frags = o.fusedg_prep(g, w, w)
            patches = o.fusedg_patches(g, "cuda")
            zf = o.fusedg_iter(g, x, z, tau, frags, -1.0, patches)
            rf = o.fusedg_assemble(g, patches, mask, x, 1.0)
            zr = o.analysis(g, x, w, -1.0, z, None, tau)
            rr = o.synthesis(g, zr, w, 1.0, None, mask, x)
            agree = (zf != 0) == (zr != 0)
            e1 = float(((zf - zr) * agree).abs().max() / zr.abs().max())
            flips = float((~agree).float().mean())
            worst["fusedg_z"] = max(worst.get("fusedg_z", 0.0), e1)
            worst["fusedg_flips"] = max(worst.get("fusedg_flips", 0.0), flips)
            if not (e1 < 3e-5 and flips < 1e-4):
                fails.append((tag, "fusedg_z", e1, flips))
            if flips == 0.0:
                e2 = rel(rf, rr)
                worst["fusedg_r"] = max(worst.get("fusedg_r", 0.0), e2)
                if not e2 < 3e-5:
                    fails.append((tag, "fusedg_r", e2))
        if verbose:
            print(json.dumps({"case": case, "geometry": tag, "fusedg": bool(o.fusedg_supported(g))}), flush=True)
    return worst, fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worst, fails = run(n, sd)
    print(json.dumps({"cases": n, "worst_rel_err": {k: float(f"{v:.3e}") for k, v in worst.items()}, "failures": fails}))
    sys.exit(1 if fails else 0)
