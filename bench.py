#!/usr/bin/env python3
"""Headline benchmark: Mpix/s denoised (fwd+bwd) at K=30, M=64, P=7 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one complete pass of the hot path over one batch: the reference's training step
(train.py:76-102) -- awgn -> forward (K unrolled iterations) -> MSE -> backward -> gradient
all-reduce (N>1) -> clip -> Adam -> project -- on a synthetic 64 x 1 x 256 x 256 batch per GPU
(BASELINE configs[1]; weak scaling: every rank owns its own 64 images).  Inputs and weights are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch is imported by the ranks only (run_rank): the launcher below must start its children before anything in
# this process can have initialised the GPU
torch = dist = None

HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: 8.0 TB/s spec
T_START = time.perf_counter()


def note(msg):
    """Progress line on stderr (the JSON result is the only thing written to stdout)."""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores(cap=16):
    """CPU threads this process may really use: affinity and cgroup quota, capped at the box's
    per-GPU CPU share (oversubscribing a shared host makes the CPU baseline meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--P", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary s2030-architecture measurement")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--precision", default="split3", choices=["split3", "bf16"],
                    help="fused-kernel arithmetic: split-bf16 x3 (fp32-grade, default) or plain bf16")
    ap.add_argument("--backend", default="auto", choices=["auto", "generic"])
    ap.add_argument("--code-storage", default="fp32", choices=["fp32", "bf16", "fp32-nchw"],
                    help="how the fused sweeps store the codes that never leave them: fp32 pixel-blocked (default, "
                         "bit-identical to the reference layout), bf16 pixel-blocked (opt-in: half the fat bytes, "
                         "PSNR parity only), or fp32 in the reference's NCHW layout")
    return ap.parse_args()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL), each
    running this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- what `python -m torch.distributed.run
    --nproc-per-node N` would do.  The parent never imports torch or touches the GPU (nothing is exec'ed from a
    process that has initialised it); rank 0's stdout (the one JSON line) is the parent's stdout, the other ranks'
    stdout goes to stderr.  Any failing rank ends the job: the rest are terminated and the parent exits non-zero."""
    import socket
    import subprocess
    assert "torch" not in sys.modules or not sys.modules["torch"].cuda.is_initialized()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's cross-process handles need it here
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    note(f"launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                note(f"launcher: rank {r} exited with {code}; stopping the others")
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def event_ms(fn, reps):
    """Average device time of `fn` over `reps` launches, HIP events on torch's current stream
    (the stream every kernel of this package is enqueued on)."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / reps


def kernel_probe(cva, net, batch, size, reps=5):
    """Time each kernel family of the step on the step's own shapes (HIP events on the launch
    stream); returns the table and the dominant one (launches-per-step x average duration) with its
    algorithmic bytes: every fat (M-channel) tensor the kernel must read or write once, plus the thin
    (one-channel) ones -- DESIGN.md section 5 lists them per kernel."""
    from cdlnet_video_amd import loop
    o = cva.ops
    K, M, P = net.K, net.M, net.P
    N, C = batch, 1
    g = o.Geometry.make(N, C, M, (size, size), (P, P), (P // 2, P // 2), 1)
    dev = "cuda"
    x = torch.randn(g.image_shape(), device=dev)
    z = torch.randn(g.code_shape(), device=dev) * (torch.rand(g.code_shape(), device=dev) < 0.2)
    gup = torch.randn(g.code_shape(), device=dev)
    tau = torch.full((N, M), 0.01, device=dev)
    w = net.A[1].weight.detach()
    out = torch.empty(g.code_shape(), device=dev)
    thin = torch.empty(g.image_shape(), device=dev)
    dt = torch.zeros(2, M, device=dev)
    fat = z.numel() * 4
    th = x.numel() * 4
    if loop.BACKEND == "auto" and o.fused_supported(g):
        prec, lay = loop.PRECISION, loop.CODE_LAYOUT
        frags = o.fused_prep(w, w)
        patches = o.fused_patches(g, dev)
        ws = o.fused_wgrad_workspace(g, dev)
        dtp = torch.empty((o.fused_tiles(g), M), device=dev)
        bits = o.fused_support_map(g, z)             # what the training forward writes next to z' (2 bits/element)
        mapb = bits.numel() * 4
        # operands in the layout the sweeps keep them in (the codes of a sweep never leave it)
        zl, gl = o.fused_from_nchw(g, z, lay), o.fused_from_nchw(g, gup, lay)
        outl = o.fused_code_buffer(g, lay, dev)[0]
        fat = outl.numel() * outl.element_size()
        table = {
            f"k_stage<FWD,{prec}> (z'=ST(z-A r), patches of B z', support map)":
                (lambda: o.fused_iter(g, x, zl, tau, frags, -1.0, patches, prec, out=outl, map_out=bits,
                                      lay_in=lay, lay_out=lay), K, 2 * fat + mapb + 2 * th),
            # the reverse sweep as cdl_fused2d_backward runs it: dA_k rides in the stage (du_k is read once, by the next
            # stage), dB_k is a single-operator filter-gradient launch
            f"k_stage<BWD,{prec}> (du=[z'!=0](du'+B^T q), dtau, patches of A^T du, dA_k)":
                (lambda: o.fused_stage_bwd(g, x, gl, bits, frags, patches, dtp, True, prec, out=outl,
                                           lay_in=lay, lay_out=lay, r2=x, alpha=-1.0, workspace=ws),
                 K, 2 * fat + mapb + 3 * th),
            f"k_wgrad2d<{prec}> (dB_k)":
                (lambda: o.fused_wgrad(g, ws, zl, x, 1.0, precision=prec, layout=lay), K, fat + th),
            "k_assemble (thin)": (lambda: o.fused_assemble(g, patches, None, x, 1.0, out=thin), 2 * K, 3 * th),
            "k_prep (weights -> bf16 fragments)": (lambda: o.fused_prep(w, w), 2 * K, 0),
        }
    else:
        table = {
            "k_analysis(fwd iter: z'=ST(z-A r))": (lambda: o.analysis(g, x, w, -1.0, z, None, tau, out=out), K, 2 * fat + th),
            "k_synthesis(fwd: r=Bz-yp)": (lambda: o.synthesis(g, z, w, 1.0, None, None, x, out=thin), K, fat + 2 * th),
            "k_analysis(bwd: g=du+B^T q)": (lambda: o.analysis(g, x, w, 1.0, gup, z, None, out=out), K, 3 * fat + th),
            "k_synthesis(bwd: q=-A^T du)": (lambda: o.synthesis(g, gup, w, -1.0, z, None, None, out=thin), K - 1, 2 * fat + th),
            "k_wgrad(dA)": (lambda: o.wgrad(g, gup, x, -1.0, gate=z), K, 2 * fat + th),
            "k_wgrad(dB)": (lambda: o.wgrad(g, z, x, 1.0), K, fat + th),
            "k_tau_partial": (lambda: o.tau_grad(g, gup, z, None, dt), K, 2 * fat),
        }
    rows = {}
    for name, (fn, count, nbytes) in table.items():
        ms = event_ms(fn, reps)
        rows[name] = {"ms": ms, "per_step": count, "bytes": nbytes,
                      "GBps": nbytes / ms / 1e6, "share_ms": ms * count}
    dom = max(rows, key=lambda k: rows[k]["share_ms"])
    return rows, dom


def kernel_source_sha():
    """Identifies the kernel revision a PMC profile was taken from."""
    import hashlib
    src = os.path.join(ROOT, "cdlnet-video_amd", "csrc", "cdl_fused2d.hip")
    return hashlib.sha1(open(src, "rb").read()).hexdigest()[:16]


def pmc_traffic(kernel_label, batch, size, M, P, layout):
    """HBM bytes per launch of the dominant kernel from a committed PMC profile of THIS kernel revision
    (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_kernels.py,
    FETCH_SIZE doubled per the gfx950 correction; the file records the sha1 of cdl_fused2d.hip and the code
    layout it was taken with).  None when no profile matches shape, layout and source: counters cannot be read
    from inside the timed process, and a profile of an older revision says nothing about this one."""
    import glob
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        sh = prof.get("shape", {})
        if (sh.get("N"), sh.get("M"), sh.get("H"), sh.get("W"), sh.get("P")) != (batch, M, size, size, P):
            continue
        if prof.get("kernel_source_sha") != sha or prof.get("layout") != layout:
            continue
        for name, row in prof.get("kernels", {}).items():
            if kernel_label.startswith(name):           # e.g. "k_stage<BWD,split3>" prefixes the probe label
                return row["traffic"], os.path.basename(path)
    return None, None


def cpu_baseline(sd, x, y, sigma, K, P, reps=2):
    """The CPU oracle (PyTorch restatement of the reference, kind='port') timed on this host's cores
    on a bounded sample of the same workload: fwd+bwd of `x.shape[0]` images."""
    from oracle import cdl_oracle as O
    torch.set_num_threads(host_cores())
    sd = dict(sd)
    sd["D.weight"] = sd["B.0.weight"]
    O.loss_and_grads(sd, x, y, K=K, P=P, s=1, sigma=sigma, adaptive=True)            # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        loss, grads, xhat = O.loss_and_grads(sd, x, y, K=K, P=P, s=1, sigma=sigma, adaptive=True)
    dt = (time.perf_counter() - t0) / reps
    pix = x.shape[0] * x.shape[2] * x.shape[3]
    return pix / dt / 1e6, xhat, loss, torch.get_num_threads()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


def run_rank(args):
    global torch, dist
    # stdout carries the ONE JSON line and nothing else: libraries underneath print there too (gloo's "[Gloo] Rank 0 is
    # connected ..." banner, RCCL with NCCL_DEBUG set), so file descriptor 1 is pointed at stderr for the whole run and
    # the line goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL ("nccl" on ROCm).  CDL_DIST_BACKEND=gloo rehearses the multi-rank
    # code path on a box with fewer GPUs than ranks (ranks then share devices round-robin).
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("CDL_DIST_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {ndev} visible "
                         f"(CDL_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)")
    dev_index = (local % ndev) if world > 1 else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import cdlnet_video_amd as cva
    from cdlnet_video_amd.parallel import GradientBucket, broadcast_parameters

    from cdlnet_video_amd import loop
    loop.set_backend(args.backend)
    loop.set_precision(args.precision)
    loop.set_code_layout({"fp32": "blocked", "bf16": "blocked_bf16", "fp32-nchw": "nchw"}[args.code_storage])
    K, M, P, B, S = args.K, args.M, args.P, args.batch, args.size
    torch.manual_seed(1)
    # the constructor prints its power-method log like the reference's; stdout carries the JSON line only
    with contextlib.redirect_stdout(sys.stderr):
        net = cva.CDLNet(K=K, M=M, P=P, s=1, C=1, t0=5e-3, adaptive=True, init=True)
    sd_cpu = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    broadcast_parameters(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    bucket = GradientBucket(net.parameters())
    if world > 1:
        bucket.attach()          # the all-reduce is queued from the end of the reverse sweep (loop.on_backward_end)

    # synthetic data, generated once and resident in HBM before timing (seeded per rank)
    gen = torch.Generator().manual_seed(1234 + rank)
    x_cpu = cva.utils.synthetic_clip((min(B, 8), 1, S, S), seed=rank)
    x_cpu = x_cpu.repeat((B + x_cpu.shape[0] - 1) // x_cpu.shape[0], 1, 1, 1)[:B]
    x = x_cpu.to(dev)
    dgen = torch.Generator(device=dev).manual_seed(99 + rank)

    def step():
        loss, _ = cva.train_step(net, opt, x, 25, clip_grad=5e-2, project=True,
                                 generator=dgen)
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"model + data ready (rank {rank}/{world}); warm-up x{args.warmup}")
    for _ in range(args.warmup):
        step()
    barrier()
    note("timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed * 1e3 / args.steps
    note(f"{ms_per_step:.1f} ms/step")
    pix_per_step = world * B * S * S
    value = pix_per_step / (ms_per_step * 1e-3) / 1e6

    # forward-only (inference) rate on the same batch, for BASELINE.md's fwd column
    with torch.no_grad():
        y_inf = x + torch.randn(x.shape, device=dev, generator=dgen) * 25 / 255
        fwd_ms = sorted(event_ms(lambda: net(y_inf, 25.0), 1) for _ in range(5))[2]      # median of 5
    fwd_mpix = B * S * S / (fwd_ms * 1e-3) / 1e6

    out = None
    syncs_per_step = (bucket.syncs / max(args.steps + args.warmup, 1)) if world > 1 else 0
    bucket.detach()          # the kernel probes below re-run steps on rank 0 ALONE: no collective may be issued there
    if rank == 0:
        note(f"fwd-only {fwd_ms:.1f} ms; probing kernels")
        rows, dom = kernel_probe(cva, net, B, S)
        # in-step durations of the fat kernels: HIP events recorded by the library around every launch of
        # the sweeps, on their stream, over further steps of the same workload (an isolated re-launch of one
        # kernel misses the cache state its predecessor leaves behind in the step, and rocprofv3 over this
        # command averages the in-step launches)
        instep = {}
        if loop.BACKEND == "auto" and any("k_stage" in k for k in rows):
            o = cva.ops
            o.fused_timing(True)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            o.fused_timing(False)
            got = o.fused_timing_read()
            for label in rows:
                cls = "stage_fwd" if label.startswith("k_stage<FWD") else "stage_bwd" if label.startswith("k_stage<BWD") \
                    else "wgrad" if label.startswith("k_wgrad2d") else None
                if cls and got[cls][1]:
                    ms = got[cls][0]
                    instep[label] = {"ms": round(ms, 4), "launches": got[cls][1],
                                     "GBps": round(rows[label]["bytes"] / ms / 1e6, 1)}
                    rows[label]["isolated_ms"] = rows[label]["ms"]
                    rows[label]["ms"] = ms
                    rows[label]["GBps"] = rows[label]["bytes"] / ms / 1e6
                    rows[label]["share_ms"] = ms * rows[label]["per_step"]
            dom = max(rows, key=lambda k: rows[k]["share_ms"])
        d = rows[dom]
        note("kernel probe done")
        traffic, traffic_src = pmc_traffic(dom, B, S, M, P, loop.CODE_LAYOUT)
        out = {
            "metric": f"Mpix/s denoised (fwd+bwd) at K={K},M={M},P={P}",
            "value": round(value, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (split-bf16 x3 MFMA, fp32 accumulate" if args.precision == "split3"
                      else "bf16 MFMA (fp32 accumulate") +
                     (", bf16 code STORAGE: opt-in, PSNR parity only)" if args.code_storage == "bf16" else ", fp32 storage)"),
            "data": "synthetic", "code_layout": loop.CODE_LAYOUT,
            "config": {"workload": f"CDLNet K={K} M={M} P={P} s=1 C=1 train step (fwd+bwd+Adam+project), "
                                   f"batch {B}x1x{S}x{S} per GPU, sigma=25",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "rccl_world": dist.get_world_size() if world > 1 else 1,
            "dist_backend": (dist.get_backend() if world > 1 else None),
            "grad_syncs_per_step": syncs_per_step,
            "fwd_only_mpix_s": round(fwd_mpix, 3), "fwd_ms": round(fwd_ms, 3),
            "loss": float(loss),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(d["GBps"], 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["GBps"] / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src, "avg_ms": round(d["ms"], 4),
                         "algorithmic_bytes_per_launch": d["bytes"]},
            "kernels": {k: dict({"ms": round(v["ms"], 4), "per_step": v["per_step"], "GBps": round(v["GBps"], 1)},
                                **({"isolated_ms": round(v["isolated_ms"], 4)} if "isolated_ms" in v else {}))
                        for k, v in rows.items()},
            "kernel_timing": "fat kernels: HIP events around every launch inside 3 further steps (in-step); "
                             "thin kernels and isolated_ms: back-to-back re-launches of one kernel",
        }
        if world == 1 and not args.no_secondary:
            # secondary line (not the metric): the SHIPPED 2-D architecture -- CDLNet-s2030: K=30 M=169 P=7 stride 2
            # (/root/reference/trained_nets/CDLNet-s2030/args.json:2-9) -- on the same batch, forward and forward + backward
            # (loss.backward(), no optimiser), through the strip kernel (cdl_strip.hip); roofline fractions from
            # BASELINE.md section 3's bytes per pixel (2 K M b / s^2 + thin)
            try:
                note("secondary: shipped s2030 architecture")
                with contextlib.redirect_stdout(sys.stderr):
                    torch.manual_seed(1)
                    net2 = cva.CDLNet(K=30, M=169, P=7, s=2, C=1, t0=5e-3, adaptive=True, init=True).to(dev)
                with torch.no_grad():
                    f2 = sorted(event_ms(lambda: net2(y_inf, 25.0), 1) for _ in range(5))[2]

                def fb2():
                    for p_ in net2.parameters():
                        p_.grad = None
                    xh, _ = net2(y_inf, 25.0)
                    torch.mean((x - xh) ** 2).backward()
                b2 = sorted(event_ms(fb2, 1) for _ in range(3))[1]
                bpp = 2 * 30 * 169 * 4 / 4 + 31 * 4
                out["secondary_s2030_arch"] = {
                    "model": "CDLNet K=30 M=169 P=7 s=2 C=1", "batch": f"{B}x1x{S}x{S}",
                    "fused": bool(cva.ops.fusedg_supported(cva.ops.Geometry.make(B, 1, 169, (S, S), (7, 7), (3, 3), 2))),
                    "fwd_ms": round(f2, 3), "fwd_mpix_s": round(B * S * S / f2 / 1e3, 2),
                    "fwd_frac_of_hbm_roofline": round(B * S * S / (f2 * 1e-3) * bpp / (HBM_PEAK_GBS * 1e9), 4),
                    "fwdbwd_ms": round(b2, 3), "fwdbwd_mpix_s": round(B * S * S / b2 / 1e3, 2),
                    "fwdbwd_frac_of_hbm_roofline": round(B * S * S / (b2 * 1e-3) * 3 * bpp / (HBM_PEAK_GBS * 1e9), 4)}
                del net2
                torch.cuda.empty_cache()
            except Exception as exc:                      # never let the secondary line cost the metric
                out["secondary_s2030_arch"] = {"error": repr(exc)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            nb = min(args.cpu_batch, B)
            xs = x_cpu[:nb]
            ys = xs + torch.randn(xs.shape, generator=gen) * 25 / 255
            note(f"CPU baseline on {host_cores()} threads")
            cpu_mpix, xref, lref, threads = cpu_baseline(sd_cpu, xs, ys, 25.0, K, P)
            note("CPU baseline done")
            with torch.no_grad():
                net0 = cva.CDLNet(K=K, M=M, P=P, s=1, C=1, t0=5e-3, adaptive=True, init=False)
                net0.load_state_dict(sd_cpu)
                xo, _ = net0.to(dev)(ys.to(dev), 25.0)
            rel = float((xo.cpu() - xref).abs().max() / xref.abs().max())
            out["cpu_baseline"] = {"value": round(cpu_mpix, 4), "unit": "Mpix/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"fwd+bwd of {nb}x1x{S}x{S} (same net, same noise model), "
                                             f"oracle/cdl_oracle.py on torch CPU, 2 reps after 1 warm-up"}
            out["parity"] = {"xhat_rel_err_vs_cpu": rel,
                             "psnr_cpu": round(cva.psnr(xs, xref), 4),
                             "psnr_gpu": round(cva.psnr(xs, xo.cpu()), 4),
                             "psnr_noisy": round(cva.psnr(xs, ys), 4)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    os.close(result_fd)


if __name__ == "__main__":
    main()
