#!/usr/bin/env python3
"""Infinity-Cache blocking experiment: run cfg2 (CDLNet K=30 M=64 P=7, 64x1x256x256) in sample chunks
small enough that one launch's fat tensors stay in the 256 MiB Infinity Cache between launches.

    python tools/bench_chunks.py [chunk sizes...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402


def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    torch.manual_seed(1)
    net = cva.CDLNet(K=30, M=64, P=7, s=1, C=1, t0=5e-3, adaptive=True, init=True).cuda()
    N = 64
    x = cva.utils.synthetic_clip((4, 1, 256, 256), seed=3).repeat(N // 4, 1, 1, 1).cuda()
    y = x + torch.randn_like(x) * 25 / 255
    pix = N * 256 * 256
    sizes = [int(a) for a in sys.argv[1:]] or [64, 32, 16, 12, 8, 6, 4]
    for n in sizes:
        def fwd():
            with torch.no_grad():
                return [net(y[i:i + n], 25.0)[0] for i in range(0, N, n)]

        def fwdbwd():
            for p in net.parameters():
                p.grad = None
            for i in range(0, N, n):
                xhat, _ = net(y[i:i + n], 25.0)
                (torch.sum((x[i:i + n] - xhat) ** 2) / x.numel()).backward()

        f, fb = ev(fwd, 3), ev(fwdbwd, 3)
        print(json.dumps({"chunk": n, "fwd_ms": round(f, 3), "fwd_mpix_s": round(pix / f / 1e3, 1),
                          "fwdbwd_ms": round(fb, 3), "fwdbwd_mpix_s": round(pix / fb / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
