"""Trainer-side restatement of the calls that drive the hot path.

`init_model` mirrors the reference's factory (train.py:180-219, train3d.py:174-206,
traincsr.py:281-302) so an `args.json` builds the same net + Adam + StepLR; `train_step` is the
body of the reference's batch loop (train.py:76-103 / train3d.py:90-118) with the data-parallel
gradient exchange added (parallel.py); `fit` is the reference's epoch driver (train.py:35-158: phases,
PSNR logs, divergence backtracking with learning-rate decay, checkpoint rotation, MC-SURE option) around
that step.  Data loading and plotting are outside the hot path and are not re-implemented here.
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn

import torch.distributed as dist

from .net import CDLNet, CDLNet_CSR, CDLNet_CSRf2, CDLNetVideo, GDLNet
from .parallel import phase_consensus
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


def mcsure_loss(net, obsrv, xhat, sigma, mask=1, h=1e-3, generator=None, b=None):
    """Monte-Carlo SURE objective of train.py:87-93: data fidelity to the OBSERVATION plus a divergence
    estimate from one extra forward at obsrv + h*b, b ~ N(0, I):
        mean((obsrv - xhat)^2) + 2 * mean((sigma/255)^2 * b * (net(obsrv + h b) - xhat)) / h
    Gradients flow through both forward passes (both run in the HIP kernels).

    The divergence term divides the difference of two forward passes by h = 1e-3, which multiplies every arithmetic
    difference between them by 1e3.  On the matrix-core paths the operands are two-term bf16 splits (16-17 significant
    bits): per-step gradients agree with fp32 autograd to ~1e-4 (tests/test_gpu_nets.py), and over several Adam steps
    weights whose SURE gradient is below that noise drift apart.  `with loop.precision_scope(
    "split4")` adds the lo * lo products on the fused 2-D path; measured, it does not change that (the operand
    truncation, not the dropped product, is the floor), so it is not selected here.  `with loop.precision_scope("fp32")`
    around the training step runs both passes and their backward on the fp32 VALU kernels (6e-6 instead of 7e-5 per step,
    several times slower); the caller chooses."""
    if b is None:
        dev = generator.device if generator is not None else obsrv.device
        b = torch.randn(obsrv.shape, device=dev, dtype=obsrv.dtype, generator=generator).to(obsrv.device)
    xhat_b, _ = net(obsrv.clone() + h * b, sigma, mask=mask)
    s2 = (sigma / 255.0) ** 2
    div = 2.0 * torch.mean(s2 * b * (xhat_b - xhat)) / h
    return torch.mean((obsrv - xhat) ** 2) + div


def train_step(net, opt, batch, noise_std, clip_grad=None, demosaic=False, project=True,
               grad_sync=None, generator=None, mcsure=False):
    """One optimiser step: awgn -> forward -> MSE (or MC-SURE) -> backward -> [all-reduce] -> clip -> Adam
    -> project (train.py:76-102).

    `grad_sync` is a callable run between backward and clipping (parallel.GradientBucket.sync).
    Returns (loss tensor, sigma).
    """
    mask = gen_bayer_mask(batch) if demosaic else 1
    noisy, sigma = awgn(batch, noise_std, generator)
    obsrv = mask * noisy
    opt.zero_grad(set_to_none=True)
    xhat, _ = net(obsrv, sigma, mask=mask)
    if mcsure:
        loss = mcsure_loss(net, obsrv, xhat, sigma, mask=mask, generator=generator)
    else:
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


# ------------------------------------------------------------------------------------------ fit loop
def grad_norm(params):
    """l2 norm of the mini-batch gradient (train.py:161-170)."""
    total = 0.0
    for p in params:
        if p.grad is not None:
            total += float(p.grad.detach().norm(2)) ** 2
    return total ** 0.5


def getlr(opt):
    return [pg["lr"] for pg in opt.param_groups]


def setlr(opt, lr):
    if not isinstance(lr, (list, tuple, np.ndarray)):
        lr = [lr for _ in opt.param_groups]
    for pg, v in zip(opt.param_groups, lr):
        pg["lr"] = float(v)


def save_args(args, ckpt=True):
    """train.py:249-258: args.json next to the checkpoints, pointing `paths.ckpt` at net.ckpt."""
    save_path = args["paths"]["save"]
    if ckpt:
        args["paths"]["ckpt"] = os.path.join(save_path, "net.ckpt")
    with open(os.path.join(save_path, "args.json"), "w") as f:
        f.write(json.dumps(args, indent=4, sort_keys=True))


def fit(net, opt, loaders, sched=None, epochs=1, device=torch.device("cpu"), save_dir=None, start_epoch=1,
        clip_grad=1, noise_std=25, demosaic=False, verbose=True, val_freq=1, save_freq=1, epoch_fun=None,
        mcsure=False, backtrack_thresh=1, grad_sync=None, log=print, group=None, generator=None):
    """The reference's training driver (train.py:35-158), same arguments and files:

    * phases train / val (every `val_freq` epochs) / test (only at `epoch == epochs`, as written there);
      val and test use the mid-range noise level;
    * per-epoch PSNR = mean over batches of -10 log10(loss), appended to `{phase}.txt`;
    * divergence backtracking: when a phase's PSNR falls `backtrack_thresh` dB below its best, or the loss
      is nan/inf, reload `net.ckpt` (`0.ckpt` while `epoch <= save_freq`), scale every learning rate by 0.8,
      rewind `epoch` to the checkpointed one and log it in `backtrack.txt`;
    * scheduler step per epoch; `net.ckpt` (+ `epoch_fun(epoch)`) every `save_freq` epochs; `0.ckpt` at start.
      Under data parallelism `epoch_fun` runs on RANK 0 ONLY, between the checkpoint write and the barrier: it is
      for rank-0 side effects (plots, copies of the checkpoint).  It must not change the net or the optimiser and
      must not issue a collective -- either would desynchronise or deadlock the replicas.
    * like the reference (train.py:113-142) the backtracking loop has NO bound: when the reloaded checkpoint
      itself cannot recover (e.g. a learning rate that diverges from `0.ckpt` at any scale the decay reaches
      within float range), `fit` repeats the epoch with lr x 0.8 indefinitely.

    `loaders` maps phase -> iterable of clean batches.  `generator` (new): the torch.Generator every noise draw uses;
    a CPU generator (torch.default_generator after torch.manual_seed) reproduces the random stream of the reference's
    CPU run on a GPU run (tests/test_gpu_trainer.py replays a reference-generated trajectory that way).  `grad_sync` (new) is run after backward for data
    parallel training (parallel.GradientBucket.sync).  Under data parallelism (an initialised process
    group; `group` selects it) every rank sees a different shard, so the decisions that change the
    training state are taken on REDUCED values: the phase PSNR is the mean over all ranks' batches and the
    nan / inf flag is the OR over ranks (parallel.phase_consensus) -- every rank backtracks, or none does.
    Rank 0 alone writes logs and checkpoints into `save_dir` (one node: the directory is shared); a
    barrier orders its writes before the other ranks' reads, and a backtrack reloads the SAME file on
    every rank, so replicas, optimiser states and learning rates stay identical.
    Returns the history [(epoch, phase, psnr)].
    """
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank0 = (not ddp) or dist.get_rank(group) == 0
    sync_dev = next(net.parameters()).device

    def barrier():
        if ddp:
            dist.barrier(group=group)

    def append(name, text):
        if rank0:
            with open(os.path.join(save_dir, name), "a") as f:
                f.write(text)

    if save_dir is None:
        raise ValueError("fit needs save_dir (checkpoints drive the backtracking)")
    os.makedirs(save_dir, exist_ok=True)
    if not isinstance(noise_std, (list, tuple)):
        noise_std = (noise_std, noise_std)
    log(f"fit: using device {device}")
    log("Saving initialization to 0.ckpt")
    if rank0:
        save_ckpt(os.path.join(save_dir, "0.ckpt"), net, 0, opt, sched)
    barrier()
    top_psnr = {"train": 0, "val": 0, "test": 0}
    history = []
    epoch = start_epoch
    while epoch < start_epoch + epochs:
        diverged, loss = False, 0.0
        for phase in ("train", "val", "test"):
            net.train() if phase == "train" else net.eval()
            if epoch != epochs and phase == "test":
                continue
            if phase == "val" and epoch % val_freq != 0:
                continue
            if phase not in loaders or loaders[phase] is None:
                continue
            phase_nstd = noise_std if phase == "train" else (noise_std[0] + noise_std[1]) / 2.0
            psnr, nb, bad = 0.0, 0, False
            for batch in loaders[phase]:
                batch = batch.to(device)
                if phase == "train":
                    loss_t, _ = train_step(net, opt, batch, phase_nstd, clip_grad=clip_grad, demosaic=demosaic,
                                           grad_sync=grad_sync, mcsure=mcsure, generator=generator)
                else:
                    with torch.no_grad():
                        mask = gen_bayer_mask(batch) if demosaic else 1
                        noisy, sigma = awgn(batch, phase_nstd, generator)
                        xhat, _ = net(mask * noisy, sigma, mask=mask)
                        loss_t = torch.mean((batch - xhat) ** 2)
                loss = float(loss_t)
                if verbose and phase == "train":
                    log(f"{phase.upper()}-E{epoch} loss={loss:.1e}|gnorm={grad_norm(net.parameters()):.1e}")
                psnr = psnr - 10 * np.log10(loss) if loss > 0 else float("nan")
                bad = bad or bool(np.isnan(loss) or np.isinf(loss))
                nb += 1
            # one reduction per phase: mean PSNR over every rank's batches, nan / inf on ANY rank
            psnr, bad = phase_consensus(psnr, nb, bad or bool(np.isnan(psnr)), sync_dev, group)
            if bad:
                psnr = float("nan")
            log(f"{phase.upper()} PSNR: {psnr:.3f} dB")
            history.append((epoch, phase, psnr))
            if psnr > top_psnr[phase]:
                top_psnr[phase] = psnr
            elif (psnr + backtrack_thresh < top_psnr[phase]) or bad:
                diverged = True
                break
            append(f"{phase}.txt", f"{psnr:.3f}, ")
        if diverged:
            ckpt_path = os.path.join(save_dir, "0.ckpt" if epoch <= save_freq else "net.ckpt")
            log(f"Loss has diverged. Backtracking to {ckpt_path} ...")
            append("backtrack.txt", f"{epoch}  ")
            epoch = epoch - save_freq if epoch % save_freq == 0 else epoch - epoch % save_freq
            old_lr = np.array(getlr(opt))
            load_ckpt(ckpt_path, net, opt, sched)
            net.to(device)
            setlr(opt, old_lr * 0.8)
            log(f"Updated Learning Rate(s): {getlr(opt)}")
            epoch = epoch + 1
            continue
        if sched is not None:
            sched.step()
        if epoch % save_freq == 0:
            if rank0:
                save_ckpt(os.path.join(save_dir, "net.ckpt"), net, epoch, opt, sched)
                if epoch_fun is not None:
                    epoch_fun(epoch)
            barrier()                            # the file is complete before any rank may reload it
        epoch = epoch + 1
    return history
