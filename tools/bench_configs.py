#!/usr/bin/env python3
"""Throughput + parity of every BASELINE.json configuration on one MI355X (secondary to bench.py,
which measures the headline configs[1]).  One JSON line per configuration:

    python tools/bench_configs.py [cfg1 cfg3 ...] > gpurun_out/configs.jsonl

For each config: forward-only and forward+backward (loss.backward(), no optimiser) Mpix/s through the
product modules, the CPU oracle on a bounded sample of the same workload (same box, host threads), the
max relative error of xhat against the oracle on that sample and both PSNRs.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import cdlnet_video_amd as cva                  # noqa: E402
from oracle import cdl_oracle as O              # noqa: E402

sys.path.insert(0, ROOT)
from bench import host_cores                    # noqa: E402

CONFIGS = {
    # name: (kind, ctor kwargs, input shape, sigma, masked, cpu sample size)
    "cfg1": ("2d", dict(K=10, M=32, P=5, s=1, C=1), (1, 1, 128, 128), 25.0, False, 1),
    "cfg1-b10": ("2d", dict(K=10, M=32, P=5, s=1, C=1), (10, 1, 128, 128), 25.0, False, 2),
    "cfg2": ("2d", dict(K=30, M=64, P=7, s=1, C=1), (64, 1, 256, 256), 25.0, False, 1),
    "cfg2-b16": ("2d", dict(K=30, M=64, P=7, s=1, C=1), (16, 1, 256, 256), 25.0, False, 1),   # cfg5's batch: per-pixel comparison
    "s2030-arch": ("2d", dict(K=30, M=169, P=7, s=2, C=1), (64, 1, 256, 256), 25.0, False, 1),
    "cfg3": ("3d", dict(K=20, M=48, P=[5, 5, 5], s=1, C=1), (8, 1, 8, 128, 128), 25.0, False, 1),
    "cfg4": ("2d", dict(K=42, M=64, P=7, s=1, C=3), (8, 3, 256, 256), (1.0, 20.0), True, 1),
    # the shipped 3-D net (/root/reference/args3dmri.json:3-14): kernel (kD, kH, kW) = (9, 9, 5), stride 2 in all three
    # directions, crops of 128 x 128, batch 1 (args3dmri.json:24-27 loads 30 frames; 16 = the model's `depth`)
    "args3dmri": ("3d", dict(K=30, M=169, P=[9, 9, 5], s=2, C=1), (1, 1, 16, 128, 128), 25.0, False, 1),
    "args3dmri-b8": ("3d", dict(K=30, M=169, P=[9, 9, 5], s=2, C=1), (8, 1, 16, 128, 128), 25.0, False, 1),
    "cfg5": ("gabor", dict(K=30, M=64, P=7, s=1, C=1, order=1, shared=""), (16, 1, 256, 256), 25.0, False, 1),
}


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


def run(name):
    kind, kw, shape, sigma, masked, ncpu = CONFIGS[name]
    torch.manual_seed(1)
    cls = {"2d": cva.CDLNet, "3d": cva.CDLNetVideo, "gabor": cva.GDLNet}[kind]
    extra = {"depth": shape[2]} if kind == "3d" else {}
    net = cls(**kw, t0=5e-3, adaptive=True, init=True, **extra)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    small = (min(shape[0], 4),) + shape[1:]
    x = cva.utils.synthetic_clip(small, seed=3)
    x = x.repeat((shape[0] + small[0] - 1) // small[0], *([1] * (len(shape) - 1)))[:shape[0]]
    gen = torch.Generator().manual_seed(4)
    y, sig = cva.awgn(x, sigma if isinstance(sigma, (tuple, list)) else float(sigma), gen)
    mask = cva.gen_bayer_mask(x) if masked else None
    if masked:
        y = mask * y
    yd, xd = y.cuda(), x.cuda()
    sd_ = sig.cuda() if torch.is_tensor(sig) else sig
    md = mask.cuda() if masked else 1
    pix = 1
    for d in (shape[0],) + tuple(shape[2:]):
        pix *= d

    def fwd():
        with torch.no_grad():
            return net(yd, sd_, mask=md)

    def fwdbwd():
        for p in net.parameters():
            p.grad = None
        xhat, _ = net(yd, sd_, mask=md)
        torch.mean((xd - xhat) ** 2).backward()

    reps = 3 if pix > 2e6 else 10
    f_ms, fb_ms = ev(fwd, reps), ev(fwdbwd, reps)
    xhat = fwd()[0].cpu()

    # CPU oracle on a bounded sample of the same workload
    torch.set_num_threads(host_cores())
    xs, ys = x[:ncpu], y[:ncpu]
    ss = sig[:ncpu] if torch.is_tensor(sig) else sig
    ms = mask[:ncpu] if masked else None
    ndim = 3 if kind == "3d" else 2
    okw = dict(K=kw["K"], P=kw["P"], s=kw["s"], sigma=ss, adaptive=True, mask=ms, ndim=ndim, gabor=kind == "gabor")
    with torch.no_grad():
        t0 = time.perf_counter(); xr, _ = O.ista(sd, ys, **okw); cf = time.perf_counter() - t0
    t0 = time.perf_counter(); O.loss_and_grads(O.gabor_alias(sd, kw["K"], "") if kind == "gabor" else sd, xs, ys, **okw)
    cfb = time.perf_counter() - t0
    cpix = pix / shape[0] * ncpu
    rel = float((xhat[:ncpu] - xr).abs().max() / xr.abs().max())
    from cdlnet_video_amd import loop, ops
    out = {"config": name, "model": f"{cls.__name__} {kw}", "input": list(shape),
           "fused_path": bool(kind in ("2d", "gabor") and kw["C"] == 1 and kw["s"] == 1 and kw["M"] in (32, 64)
                              and kw["P"] <= 7),
           "fwd_ms": round(f_ms, 3), "fwd_mpix_s": round(pix / f_ms / 1e3, 3),
           "fwdbwd_ms": round(fb_ms, 3), "fwdbwd_mpix_s": round(pix / fb_ms / 1e3, 3),
           "cpu_fwd_mpix_s": round(cpix / cf / 1e6, 4), "cpu_fwdbwd_mpix_s": round(cpix / cfb / 1e6, 4),
           "cpu_threads": torch.get_num_threads(), "cpu_sample": ncpu,
           "xhat_rel_err": rel, "psnr_cpu": round(O.psnr(xs, xr), 4), "psnr_gpu": round(O.psnr(xs, xhat[:ncpu]), 4),
           "psnr_noisy": round(O.psnr(xs, ys if not masked else xr * 0 + ys), 3)}
    print(json.dumps(out), flush=True)


def run_csr(name):
    """argscsr.json: CDLNet_CSRf2 K=30 M=169 P=9 s=2.  Inference = analyzemri.py:162-182 (two network
    calls per frame) over a T-frame clip; training = the five-call chain of traincsr.py:257-261 on
    128x128 crops, batch 1, loss.backward()."""
    torch.manual_seed(1)
    kw = dict(K=30, M=169, P=9, s=2, C=1)
    net = cva.CDLNet_CSRf2(**kw, t0=5e-3, adaptive=True, init=True)
    with torch.no_grad():
        net.g1.fill_(0.5)
        net.g2.fill_(0.5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    T, H, W = 8, 256, 256
    vid = cva.utils.synthetic_clip((1, 1, T, H, W), seed=3)
    gen = torch.Generator().manual_seed(4)
    frames = [vid[:, :, t] + torch.randn(1, 1, H, W, generator=gen) * 25 / 255 for t in range(T)]
    fd = [f.cuda() for f in frames]
    inf_ms = ev(lambda: cva.csr_inference_v2(net, fd, 25.0), 3)
    out = [o.cpu() for o in cva.csr_inference_v2(net, fd, 25.0)]

    crops = [f[..., :128, :128].contiguous() for f in fd[:3]]
    clean = [vid[:, :, t, :128, :128].cuda() for t in range(3)]
    mse = lambda a, b: torch.mean((a - b) ** 2)

    def chain():
        for p in net.parameters():
            p.grad = None
        xp, zp = net(crops[0], None, None, 25.0)
        xc, zc = net(crops[1], zp, None, 25.0)
        xa, za = net(crops[2], zc, None, 25.0)
        xc2, _ = net(crops[1], zp, za, 25.0)
        xp2, _ = net(crops[0], None, za, 25.0)
        (mse(clean[0], xp) + mse(clean[1], xc) + mse(clean[2], xa) + mse(clean[1], xc2) + mse(clean[0], xp2)).backward()

    trn_ms = ev(chain, 3)

    torch.set_num_threads(host_cores())
    okw = dict(K=30, P=9, s=2, sigma=25.0, adaptive=True, variant="f2")
    Tc = 2
    with torch.no_grad():
        t0 = time.perf_counter()
        codes = [None] * (Tc + 2)
        for t in range(Tc):
            _, codes[t + 1] = O.ista_csr(sd, frames[t], codes[t], None, **okw)
        ref = [O.ista_csr(sd, frames[t], codes[t], codes[t + 1], **okw)[0] for t in range(Tc)]
        cpu_s = time.perf_counter() - t0
    got2 = [o.cpu() for o in cva.csr_inference_v2(net, fd[:Tc], 25.0)]
    rel = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(got2, ref))
    print(json.dumps({
        "config": name, "model": f"CDLNet_CSRf2 {kw}", "clip": [T, H, W],
        "inference": "csr_inference_v2 (2 network calls per frame)", "fused_path": False,
        "infer_ms_per_clip": round(inf_ms, 2), "infer_frames_s": round(T / inf_ms * 1e3, 2),
        "infer_mpix_s": round(T * H * W / inf_ms / 1e3, 3),
        "train_chain_ms": round(trn_ms, 2), "train_chain": "5 calls, 128x128 crops, batch 1, fwd+bwd",
        "cpu_infer_mpix_s": round(Tc * H * W / cpu_s / 1e6, 4), "cpu_threads": torch.get_num_threads(),
        "cpu_sample": f"{Tc} frames", "xhat_rel_err": rel,
        "psnr_cpu": round(O.psnr(vid[:, :, 1], ref[1]), 4), "psnr_gpu": round(O.psnr(vid[:, :, 1], got2[1]), 4),
        "psnr_noisy": round(O.psnr(vid[:, :, 1], frames[1]), 3)}), flush=True)


if __name__ == "__main__":
    for cfg in (sys.argv[1:] or list(CONFIGS) + ["csr-f2"]):
        print(f"[{cfg}] ...", file=sys.stderr, flush=True)
        run_csr(cfg) if cfg == "csr-f2" else run(cfg)
