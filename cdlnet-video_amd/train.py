"""Trainer-side restatement of the calls that drive the hot path.

`init_model` mirrors the reference's factory (train.py:180-219, train3d.py:174-206,
traincsr.py:281-302) so an `args.json` builds the same net + Adam + StepLR; `train_step` is the
body of the reference's batch loop (train.py:76-103 / train3d.py:90-118) with the data-parallel
gradient exchange added (parallel.py).  Data loading, checkpoint rotation, backtracking and
plotting are outside the hot path and are not re-implemented here.
"""
import torch
import torch.nn as nn

from .net import CDLNet, CDLNet_CSR, CDLNet_CSRf2, CDLNetVideo, GDLNet
from .utils import awgn, gen_bayer_mask

MODEL_TYPES = {
    "CDLNet": CDLNet,
    "JDD_CDLNet": CDLNet,          # BASELINE config 4: CDLNet with C=3 + Bayer mask
    "CDLNetVideo": CDLNetVideo,
    "GDLNet": GDLNet,
    "CDLNet_CSR": CDLNet_CSR,      # traincsr.py:290-296
    "CDLNet_CSRf2": CDLNet_CSRf2,
}


def build_model(args, init=None):
    """`Cls(**args['model'])` selected by args['type'] (train.py:184-196)."""
    model_type, model_args = args["type"], dict(args["model"])
    if model_type not in MODEL_TYPES:
        raise NotImplementedError(model_type)
    if init is None:
        init = not (args.get("paths", {}) or {}).get("ckpt")
    model_args.setdefault("init", init)
    if "init" in args["model"] and model_type != "CDLNetVideo":
        model_args["init"] = init
    return MODEL_TYPES[model_type](**model_args)


def load_ckpt(path, net, opt=None, sched=None):
    """train.py:232-247: tolerant of missing optimizer/scheduler state and of a missing `g`."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(ckpt["net_state_dict"])
    epoch = ckpt.get("epoch", 0)
    if opt is not None and "opt_state_dict" in ckpt:
        opt.load_state_dict(ckpt["opt_state_dict"])
    if sched is not None and "sched_state_dict" in ckpt:
        sched.load_state_dict(ckpt["sched_state_dict"])
    return net, opt, sched, epoch


def save_ckpt(path, net, epoch=None, opt=None, sched=None):
    """train.py:221-230: same dictionary layout."""
    torch.save({"epoch": epoch, "net_state_dict": net.state_dict(),
                "opt_state_dict": None if opt is None else opt.state_dict(),
                "sched_state_dict": None if sched is None else sched.state_dict()}, path)


def init_model(args, device=torch.device("cpu")):
    """Returns (net, opt, sched, epoch0) like the reference's init_model."""
    net = build_model(args)
    net.to(device)
    train_args = args.get("train", {})
    opt = torch.optim.Adam(net.parameters(), **train_args.get("opt", {"lr": 1e-3}))
    sched = torch.optim.lr_scheduler.StepLR(opt, **train_args.get("sched", {"step_size": 50, "gamma": 0.95}))
    epoch0 = 0
    ckpt = (args.get("paths", {}) or {}).get("ckpt")
    if ckpt:
        net, opt, sched, epoch0 = load_ckpt(ckpt, net, opt, sched)
        net.to(device)
    return net, opt, sched, epoch0


def train_step(net, opt, batch, noise_std, clip_grad=None, demosaic=False, project=True,
               grad_sync=None, generator=None):
    """One optimiser step: awgn -> forward -> MSE -> backward -> [all-reduce] -> clip -> Adam -> project.

    `grad_sync` is a callable run between backward and clipping (parallel.GradientBucket.sync).
    Returns (loss tensor, sigma).
    """
    mask = gen_bayer_mask(batch) if demosaic else 1
    noisy, sigma = awgn(batch, noise_std, generator)
    obsrv = mask * noisy
    opt.zero_grad(set_to_none=True)
    xhat, _ = net(obsrv, sigma, mask=mask)
    loss = torch.mean((batch - xhat) ** 2)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    if clip_grad is not None:
        nn.utils.clip_grad_norm_(net.parameters(), clip_grad)
    opt.step()
    if project and hasattr(net, "project"):
        net.project()
    return loss.detach(), sigma
