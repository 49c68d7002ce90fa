"""Data-parallel plumbing on CPU: world_size-2 gloo processes (the N>1 path of bench.py / train_step)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cdlnet_video_amd as cva
from cdlnet_video_amd.parallel import (GradientBucket, all_reduce_scalar, broadcast_parameters,
                                       shard_batch)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # different weights per rank before the broadcast
        net = cva.CDLNet(K=2, M=4, P=5, s=1, C=1, t0=1e-2, adaptive=True, init=False)
        broadcast_parameters(net, src=0)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same_after_bcast = all(torch.equal(gathered[0], g) for g in gathered)

        # fake per-rank gradients: rank r contributes (r+1) * ones; `g` gets none (as in the real net)
        bucket = GradientBucket(net.parameters())
        for name, p in net.named_parameters():
            p.grad = None if name == "g" else torch.full_like(p, float(rank + 1))
        bucket.sync()
        expect = sum(range(1, world + 1)) / world
        ok_mean = all(torch.allclose(p.grad, torch.full_like(p, expect)) for n, p in net.named_parameters() if n != "g")
        ok_g = torch.equal(net.g.grad, torch.zeros_like(net.g))
        views = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))

        # an optimiser step on identical averaged grads keeps the replicas identical
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        dist.all_gather(gathered, flat)
        same_after_step = all(torch.equal(gathered[0], g) for g in gathered)

        batch = torch.arange(7 * 3, dtype=torch.float32).reshape(7, 3)
        mine = shard_batch(batch)
        sizes = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([float(mine.shape[0])]))
        loss_mean = float(all_reduce_scalar(torch.tensor(float(rank)), "mean"))
        if rank == 0:
            out.put(dict(same_after_bcast=same_after_bcast, ok_mean=ok_mean, ok_g=ok_g, views=views,
                         same_after_step=same_after_step, sizes=[int(s) for s in sizes],
                         first_row=float(mine[0, 0]), loss_mean=loss_mean))
    finally:
        dist.destroy_process_group()


def test_gradient_bucket_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["same_after_bcast"] and res["ok_mean"] and res["ok_g"] and res["views"]
    assert res["same_after_step"]
    assert res["sizes"] == [4, 3] and res["first_row"] == 0.0          # ragged tail goes to the low ranks
    assert abs(res["loss_mean"] - 0.5) < 1e-6


def test_shard_batch_single_process():
    x = torch.arange(10).reshape(10, 1)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert [p.shape[0] for p in parts] == [3, 3, 2, 2]
    assert torch.equal(torch.cat(parts), x)
    assert shard_batch(x).shape[0] == 10                     # no process group: the whole batch
    b = GradientBucket([torch.nn.Parameter(torch.ones(3))])
    b.sync()                                                 # no process group: a no-op that still adopts
    assert b.params[0].grad is not None


# ---- fit() under data parallelism: only ONE rank's shard diverges ------------------------------------------
class _Scale(torch.nn.Module):
    """xhat = w * y (CPU stand-in for the HIP nets: fit only calls net(obsrv, sigma, mask=mask));
    `poison_at` blows the output up on one chosen training call."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.5))
        self.calls, self.poison_at = 0, None

    def forward(self, y, sigma=None, mask=1):
        out = self.w * y
        if self.training:
            self.calls += 1
            if self.poison_at is not None and self.calls == self.poison_at:
                self.poison_at = None
                out = out + 30.0
        return out, None


def _fit_worker(rank, world, port, save_dir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cdlnet_video_amd import train as T
        torch.manual_seed(0)
        net = _Scale()
        if rank == 1:
            net.poison_at = 4                           # first batch of epoch 2, on rank 1 ONLY
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        bucket = GradientBucket(net.parameters())
        g = torch.Generator().manual_seed(10 + rank)    # every rank sees its own shard
        loaders = {"train": [torch.rand(2, 1, 8, 8, generator=g) for _ in range(3)]}
        hist = T.fit(net, opt, loaders, epochs=3, save_dir=save_dir, noise_std=25, val_freq=10, save_freq=1,
                     verbose=False, backtrack_thresh=1, grad_sync=bucket.sync, log=lambda *_: None)
        state = torch.tensor([float(net.w.detach()), T.getlr(opt)[0], float(len(hist))], dtype=torch.float64)
        gathered = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        if rank == 0:
            out.put(dict(states=[t.tolist() for t in gathered], epochs=[e for e, p, _ in hist if p == "train"],
                         psnrs=[v for _, _, v in hist]))
    finally:
        dist.destroy_process_group()


def test_fit_backtracks_on_every_rank_when_one_shard_diverges(tmp_path):
    """decisions that change the training state are taken on reduced values, so a divergence seen
    by one rank rewinds ALL replicas: weights, learning rates and epoch counters stay identical."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res["states"]
    assert s0 == s1, (s0, s1)                                   # identical weight, LR, history length
    assert res["epochs"] == [1, 2, 2, 3]                        # epoch 2 repeated after the joint reload
    assert abs(s0[1] - 0.8e-2) < 1e-12                          # 0.8x once, on both ranks
    assert open(tmp_path / "backtrack.txt").read().split() == ["2"]       # written once (rank 0)
    assert len(open(tmp_path / "train.txt").read().split(",")) == 4


def test_adopt_moves_adjacent_gradients_with_one_copy():
    """The reverse sweep returns dA_0..dB_{K-1} as views of one allocation: adopt() must not copy per parameter."""
    params = [torch.nn.Parameter(torch.zeros(2, 3)) for _ in range(5)] + [torch.nn.Parameter(torch.zeros(4))]
    b = GradientBucket(params)
    block = torch.arange(30, dtype=torch.float32)
    for i in range(5):
        params[i].grad = block[6 * i:6 * (i + 1)].view(2, 3)
    params[5].grad = torch.full((4,), 7.0)
    b.adopt()
    assert b.copies == 2
    assert torch.equal(b.flat[:30], block) and torch.equal(b.flat[30:], torch.full((4,), 7.0))
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b.params, b.views))
    b.adopt()
    assert b.copies == 0                                        # already in place
