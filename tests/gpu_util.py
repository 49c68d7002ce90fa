"""Shared helpers for the -m gpu parity tests (they call the product through its C ABI)."""
import os

import torch

from conftest import ROOT, load_golden, rel_err   # noqa: F401

LOG = os.path.join(ROOT, "gpurun_out", "parity_log.txt")


def log(line):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(line + "\n")


def check(name, got, ref, tol):
    """Assert max|got-ref|/max|ref| < tol and record the measured value."""
    err = rel_err(got.detach().float().cpu(), ref.detach().float().cpu())
    log(f"{name:60s} rel_err={err:.3e} tol={tol:.1e}")
    assert err < tol, f"{name}: rel_err {err:.3e} >= {tol:.1e}"
    return err


def hyper(g):
    K, M, P, s, C = g["hyper"]
    P = tuple(g["P3"]) if "P3" in g else P
    return K, M, P, s, C


def build_from_golden(g, kind, **extra):
    """Construct the product module for a fixture and load the reference's state_dict into it."""
    import cdlnet_video_amd as cva
    K, M, P, s, C = hyper(g)
    if kind == "2d":
        net = cva.CDLNet(K=K, M=M, P=P, s=s, C=C, t0=0.0, adaptive=True, init=False)
    elif kind == "3d":
        net = cva.CDLNetVideo(K=K, M=M, P=list(P), s=s, C=C, t0=0.0, adaptive=True, init=False)
    else:
        net = cva.GDLNet(K=K, M=M, P=P, s=s, C=C, t0=0.0, order=g["order"], shared=g["shared"],
                         init=False, **extra)
    net.load_state_dict(g["sd"])
    return net.cuda()
