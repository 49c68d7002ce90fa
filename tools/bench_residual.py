#!/usr/bin/env python3
"""ResidualBlock / CDLNetVideo(residual=True) timings on one MI355X.

    python tools/bench_residual.py > gpurun_out/residual.jsonl

One JSON line per case: the block alone at M = 64 on a 16 x 128 x 128 code (forward, forward+backward, the
fp32-equivalent TFLOP/s = 2*M*M*27 flops per voxel per convolution, and the MFMA-issued rate = 3x that for the
split-bf16 products), and the cfg3 architecture with residual blocks (forward / forward+backward per clip, parity
and CPU time of the oracle on a bounded sample).
"""
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402
from oracle import cdl_oracle as O              # noqa: E402
from bench import host_cores                    # noqa: E402


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


def block_case(M, shape, reps=10):
    gen = torch.Generator().manual_seed(7)
    x = (torch.randn((1, M) + shape, generator=gen) * 0.5).cuda()
    w1 = (torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5).cuda()
    w2 = (torch.randn((M, M, 3, 3, 3), generator=gen) / (27 * M) ** 0.5).cuda()
    gout = torch.randn((1, M) + shape, generator=gen).cuda()
    g = cva.ops.residual_geometry(x, w1)
    h, out = cva.ops.residual_forward(g, x, w1, w2)
    t_f = ev(lambda: cva.ops.residual_forward(g, x, w1, w2), reps)
    t_b = ev(lambda: cva.ops.residual_backward(g, x, h, out, w1, w2, gout), reps)
    vox = shape[0] * shape[1] * shape[2]
    conv = 2.0 * M * M * 27 * vox                       # flops of one convolution
    rec = {"case": f"ResidualBlock M={M} code {shape}", "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4),
           "fwd_tflops_fp32_equiv": round(2 * conv / t_f / 1e9, 1), "fwd_tflops_mfma_issued": round(6 * conv / t_f / 1e9, 1),
           "bwd_tflops_fp32_equiv": round(4 * conv / t_b / 1e9, 1), "bwd_tflops_mfma_issued": round(12 * conv / t_b / 1e9, 1),
           "mfma_peak_tflops_bf16": 2500.0}
    print(json.dumps(rec), flush=True)


def net_case(kw, shape, ncpu=1):
    torch.manual_seed(1)
    with contextlib.redirect_stdout(sys.stderr):         # the constructor reports its power method
        net = cva.CDLNetVideo(**kw, t0=5e-3, adaptive=True, init=True, depth=shape[2], residual=True)
    with torch.no_grad():                                # blocks at a scale that keeps the codes O(1)
        for b in net.residual_blocks:
            b.conv1.weight.mul_(0.5)
            b.conv2.weight.mul_(0.5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = cva.utils.synthetic_clip(shape, seed=3)
    y, _ = cva.awgn(x, 25.0, torch.Generator().manual_seed(4))
    yd, xd = y.cuda(), x.cuda()

    def fwd():
        with torch.no_grad():
            return net(yd, 25.0)

    def fwdbwd():
        for p in net.parameters():
            p.grad = None
        xhat, _ = net(yd, 25.0)
        torch.mean((xd - xhat) ** 2).backward()

    t_f, t_fb = ev(fwd, 3), ev(fwdbwd, 3)
    xhat = fwd()[0].cpu()
    torch.set_num_threads(host_cores())
    t0 = time.perf_counter()
    ref, _ = O.ista_video_residual(sd, y[:ncpu], K=kw["K"], P=tuple(kw["P"]), s=kw["s"], sigma=25.0, adaptive=True)
    t_cpu = time.perf_counter() - t0
    err = float((xhat[:ncpu] - ref).abs().max() / ref.abs().max())
    pix = shape[0] * shape[2] * shape[3] * shape[4]
    rec = {"case": f"CDLNetVideo residual=True {kw} clip {shape}", "fwd_ms": round(t_f, 3), "fwdbwd_ms": round(t_fb, 3),
           "fwd_Mpix_s": round(pix / t_f / 1e3, 2), "fwdbwd_Mpix_s": round(pix / t_fb / 1e3, 2),
           "xhat_rel_err_vs_oracle": err, "psnr_product": round(float(O.psnr(x[:ncpu], xhat[:ncpu])), 3),
           "psnr_oracle": round(float(O.psnr(x[:ncpu], ref)), 3),
           "cpu_oracle_s_per_sample": round(t_cpu / ncpu, 2), "cpu_threads": host_cores()}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    block_case(64, (16, 128, 128))
    block_case(32, (16, 128, 128))
    net_case(dict(K=20, M=48, P=[5, 5, 5], s=1, C=1), (2, 1, 8, 128, 128))
