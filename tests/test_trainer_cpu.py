"""Control flow of the reference's epoch driver (train.py:35-158) as restated in cdlnet_video_amd.train.fit:
phases, PSNR logs, checkpoint rotation, divergence backtracking with learning-rate decay, MC-SURE loss.
`fit` only calls `net(obsrv, sigma, mask=mask)`, so a plain CPU module stands in for the HIP nets here."""
import json
import os

import torch
import torch.nn as nn

import cdlnet_video_amd as cva
from cdlnet_video_amd import train as T


class Scale(nn.Module):
    """xhat = w * y; `poison` multiplies the output on one chosen training call (a loss blow-up)."""

    def __init__(self, w=0.5):
        super().__init__()
        self.w = nn.Parameter(torch.tensor(float(w)))
        self.calls, self.poison_at, self.projected = 0, None, 0

    def forward(self, y, sigma=None, mask=1):
        out = self.w * y
        if self.training:
            self.calls += 1
            if self.poison_at is not None and self.calls == self.poison_at:
                self.poison_at = None
                out = out + 30.0
        return out, None

    def project(self):
        self.projected += 1


def loaders(n_train=3):
    g = torch.Generator().manual_seed(0)
    mk = lambda n: [torch.rand(2, 1, 8, 8, generator=g) for _ in range(n)]
    return {"train": mk(n_train), "val": mk(1), "test": mk(1)}


def test_fit_runs_phases_and_rotates_checkpoints(tmp_path):
    torch.manual_seed(0)
    net = Scale()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    seen = []
    hist = T.fit(net, opt, loaders(), sched=sched, epochs=3, save_dir=str(tmp_path), noise_std=(20, 30),
                 val_freq=2, save_freq=1, verbose=False, epoch_fun=seen.append, log=lambda *_: None)
    phases = [(e, p) for e, p, _ in hist]
    assert phases == [(1, "train"), (2, "train"), (2, "val"), (3, "train"), (3, "test")]     # test only at epoch == epochs
    assert seen == [1, 2, 3] and net.projected == 9
    assert set(os.listdir(tmp_path)) >= {"0.ckpt", "net.ckpt", "train.txt", "val.txt", "test.txt"}
    assert len(open(tmp_path / "train.txt").read().split(",")) == 4                           # "a, b, c, "
    ck = torch.load(tmp_path / "net.ckpt", weights_only=True)
    assert ck["epoch"] == 3 and set(ck) == {"epoch", "net_state_dict", "opt_state_dict", "sched_state_dict"}
    assert abs(T.getlr(opt)[0] - 1e-2 * 0.5 ** 3) < 1e-12
    assert torch.load(tmp_path / "0.ckpt", weights_only=True)["epoch"] == 0


def test_fit_backtracks_on_divergence(tmp_path):
    torch.manual_seed(0)
    net = Scale()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    net.poison_at = 4                                  # first batch of epoch 2: PSNR collapses
    logs = []
    hist = T.fit(net, opt, loaders(), epochs=3, save_dir=str(tmp_path), noise_std=25, val_freq=10,
                 save_freq=1, verbose=False, backtrack_thresh=1, log=logs.append)
    epochs_run = [e for e, p, _ in hist if p == "train"]
    assert epochs_run == [1, 2, 2, 3]                                        # epoch 2 is repeated after the reload
    assert open(tmp_path / "backtrack.txt").read().split() == ["2"]
    assert abs(T.getlr(opt)[0] - 0.8e-2) < 1e-12                             # 0.8x on every backtrack
    assert any("Backtracking to" in str(m) and "net.ckpt" in str(m) for m in logs)
    assert len(open(tmp_path / "train.txt").read().split(",")) == 4          # the diverged epoch is not logged
    bad, good = hist[1][2], hist[2][2]
    assert good > bad + 5


def test_fit_backtracks_to_initialisation_early(tmp_path):
    net = Scale()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    net.poison_at = 4                                  # epoch 2 with save_freq = 5: only 0.ckpt exists
    w0 = float(net.w.detach())
    logs = []
    T.fit(net, opt, loaders(), epochs=2, save_dir=str(tmp_path), noise_std=25, val_freq=10, save_freq=5,
          verbose=False, log=logs.append)
    assert any("0.ckpt" in str(m) for m in logs)
    assert not os.path.exists(tmp_path / "net.ckpt")
    assert float(net.w.detach()) != w0                       # trained on after the reload


def test_mcsure_loss_formula():
    net = Scale(0.7)
    g = torch.Generator().manual_seed(3)
    y = torch.rand(2, 1, 6, 6, generator=g)
    b = torch.randn(y.shape, generator=g)
    sigma = torch.tensor([20.0, 30.0]).reshape(2, 1, 1, 1)
    xhat, _ = net(y, sigma)
    got = cva.mcsure_loss(net, y, xhat, sigma, b=b)
    want = torch.mean((y - xhat) ** 2) + 2.0 * torch.mean((sigma / 255) ** 2 * b * (0.7 * 1e-3 * b)) / 1e-3
    assert abs(float(got) - float(want)) < 1e-6
    got.backward()
    assert net.w.grad is not None


def test_save_args_round_trip(tmp_path):
    args = {"type": "CDLNet", "model": {"K": 2, "M": 4, "P": 5, "s": 1, "C": 1, "adaptive": True},
            "paths": {"save": str(tmp_path), "ckpt": None}, "train": {"opt": {"lr": 1e-3},
                                                                      "sched": {"step_size": 5, "gamma": 0.9}}}
    net, opt, sched, e0 = cva.init_model(args)
    assert e0 == 0
    cva.save_ckpt(os.path.join(str(tmp_path), "net.ckpt"), net, 4, opt, sched)
    cva.save_args(args, True)
    again = json.load(open(tmp_path / "args.json"))
    assert again["paths"]["ckpt"].endswith("net.ckpt")
    net2, _, _, e1 = cva.init_model(again)             # ckpt present -> no power method, weights restored
    assert e1 == 4 and torch.equal(net2.A[1].weight, net.A[1].weight)
