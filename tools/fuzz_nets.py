#!/usr/bin/env python3
"""GPU-box: seeded random small nets (CDLNet 2-D with C in {1, 3} + mask, CDLNetVideo) on random ragged input sizes,
forward + backward through the fused sweeps (backend "auto") against the generic three-launch sweeps (backend "generic",
pinned to the oracle by the -m gpu suite): xhat and every parameter gradient.  The two paths share no kernel on the fat
tensors, so an indexing slip in either shows up; soft-threshold support flips between them are rare and small.

    python tools/fuzz_nets.py [cases] [seed]
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst_x, worst_g, fails = 0.0, 0.0, []
for case in range(cases):
    kind = rng.choice(["2d", "2d", "jdd", "3d"])
    K = rng.randint(2, 4)
    if kind == "3d":
        M, P = rng.choice([16, 24, 48]), [rng.choice([3, 5]), rng.choice([3, 5]), 0]
        P[2] = P[1]
        shape = (rng.randint(1, 3), 1, rng.randint(3, 8), rng.randint(17, 50), rng.randint(33, 90))
        torch.manual_seed(case)
        net = cva.CDLNetVideo(K=K, M=M, P=P, s=1, C=1, t0=5e-3, adaptive=True, depth=shape[2], init=True).cuda()
        masked = False
    else:
        C = 3 if kind == "jdd" else 1
        M, P = rng.choice([32, 64] if C == 1 else [16, 40, 64]), rng.choice([3, 5, 7])
        shape = (rng.randint(1, 4), C, rng.randint(20, 70), rng.randint(33, 140))
        torch.manual_seed(case)
        net = cva.CDLNet(K=K, M=M, P=P, s=1, C=C, t0=5e-3, adaptive=True, init=True).cuda()
        masked = kind == "jdd"
    x = cva.utils.synthetic_clip(shape, seed=case).cuda()
    torch.manual_seed(100 + case)
    y, sigma = cva.utils.awgn(x, (15, 35))
    mask = cva.utils.gen_bayer_mask(x) if masked else 1
    out = {}
    for backend in ("auto", "generic"):
        cva.loop.set_backend(backend)
        net.zero_grad(set_to_none=True)
        xhat, _ = net(mask * y, sigma, mask=mask)
        loss = torch.mean((x - xhat) ** 2)
        loss.backward()
        out[backend] = (xhat.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    cva.loop.set_backend("auto")
    ex = float((out["auto"][0] - out["generic"][0]).abs().max() / out["generic"][0].abs().max())
    eg = max(float((out["auto"][1][n] - g).abs().max() / g.abs().max().clamp_min(1e-20)) for n, g in out["generic"][1].items())
    worst_x, worst_g = max(worst_x, ex), max(worst_g, eg)
    tag = f"{kind} K{K} M{M} P{P} {shape}"
    if not (ex < 1e-5 and eg < 5e-3):
        fails.append((tag, ex, eg))
    print(json.dumps({"case": case, "net": tag, "xhat": float(f"{ex:.2e}"), "grad": float(f"{eg:.2e}")}), flush=True)
print(json.dumps({"cases": cases, "worst_xhat": float(f"{worst_x:.3e}"), "worst_grad": float(f"{worst_g:.3e}"), "failures": fails}))
sys.exit(1 if fails else 0)
