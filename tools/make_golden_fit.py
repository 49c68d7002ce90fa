#!/usr/bin/env python3
"""Generate tests/golden/f12_fit_trajectory.npz: the UNMODIFIED reference epoch driver (train.py:35-158 `fit`) run on
CPU around the unmodified reference CDLNet, on synthetic batches, through a forced divergence + backtrack and through
an MC-SURE run.  Build container only (needs /root/reference).

Shims: the empty `torchvision` stub of tools/make_golden.py (utils.py:4, data.py:7-9 import a package that is not
installed and that `fit` never calls).  Nothing of the reference is copied: the fixture holds inputs (initial
state_dict, batches, hyper-parameters, seed) and outputs (the PSNR log files' text, backtrack.txt, learning rates,
final state_dict).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_fit.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from make_golden import import_reference, OUT      # noqa: E402

MODEL = dict(K=3, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True)
SHAPE = (2, 1, 40, 72)


def batches(n, seed):
    import cdlnet_video_amd as cva
    return [cva.utils.synthetic_clip(SHAPE, seed=seed + i) for i in range(n)]


def read(d, name):
    p = os.path.join(d, name)
    return open(p).read() if os.path.exists(p) else ""


def run(ref_train, net_mod, sd0, loaders, seed, blow_up, **kw):
    net = net_mod.CDLNet(**MODEL, init=False)
    net.load_state_dict(sd0)
    opt = torch.optim.Adam(net.parameters(), lr=kw.pop("lr"))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.9)
    fired = []

    def epoch_fun(epoch):                       # runs right after net.ckpt of `epoch` is written (train.py:150-154)
        if blow_up and epoch == 1 and not fired:
            fired.append(1)
            with torch.no_grad():
                net.B[0].weight.mul_(1e3)

    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()) as out, \
            contextlib.redirect_stderr(io.StringIO()):
        torch.manual_seed(seed)
        ref_train.fit(net, opt, loaders, sched=sched, save_dir=d, device=torch.device("cpu"), verbose=False,
                      epoch_fun=epoch_fun, **kw)
        files = {n: read(d, n) for n in ("train.txt", "val.txt", "test.txt", "backtrack.txt")}
    text = out.getvalue()
    return net, opt, files, text


def main():
    net_mod, _ = import_reference()
    import train as ref_train
    torch.manual_seed(7)
    net0 = net_mod.CDLNet(**MODEL, init=True)
    sd0 = {k: v.clone() for k, v in net0.state_dict().items()}
    loaders = {"train": batches(3, 0), "val": batches(1, 50), "test": batches(1, 60)}
    arrays = {"shape": np.array(SHAPE)}
    for k, v in sd0.items():
        arrays["init/" + k] = v.numpy()
    for ph, bs in loaders.items():
        arrays["data/" + ph] = torch.stack(bs).numpy()
    runs = {
        "bt": dict(seed=123, blow_up=True, lr=2e-3, epochs=3, clip_grad=5e-2, noise_std=(20, 30), val_freq=1,
                   save_freq=1, backtrack_thresh=1, mcsure=False),
        "sure": dict(seed=321, blow_up=False, lr=1e-3, epochs=2, clip_grad=5e-2, noise_std=25, val_freq=1,
                     save_freq=1, backtrack_thresh=1, mcsure=True),
    }
    for tag, kw in runs.items():
        kw = dict(kw)
        net, opt, files, text = run(ref_train, net_mod, sd0, loaders, kw.pop("seed"), kw.pop("blow_up"), **kw)
        for n, t in files.items():
            arrays[f"{tag}/{n}"] = np.array(t)
        arrays[f"{tag}/lr"] = np.array([pg["lr"] for pg in opt.param_groups])
        arrays[f"{tag}/n_backtracks"] = np.array(text.count("Backtracking"))
        for k, v in net.state_dict().items():
            arrays[f"{tag}/final/" + k] = v.numpy()
        print(tag, files, [pg["lr"] for pg in opt.param_groups], text.count("Backtracking"))
    path = os.path.join(OUT, "f12_fit_trajectory.npz")
    np.savez_compressed(path, **arrays)
    print(f"f12_fit_trajectory: {os.path.getsize(path)/1024:.1f} KiB")


if __name__ == "__main__":
    main()
