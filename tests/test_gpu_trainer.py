"""The reference's epoch driver (train.py:35-158, restated in cdlnet_video_amd.train.fit) around REAL nets: every
forward / backward of the run goes through the HIP kernels (tests/test_trainer_cpu.py covers the control flow with a
CPU stand-in module)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _loaders(shape, n_train, seed):
    import cdlnet_video_amd as cva
    mk = lambda n, s: [cva.utils.synthetic_clip(shape, seed=s + i).cuda() for i in range(n)]
    return {"train": mk(n_train, seed), "val": mk(1, seed + 50), "test": mk(1, seed + 60)}


def test_fit_trains_a_cdlnet_on_the_gpu(tmp_path):
    """2-D net on the fused MFMA path: 3 epochs of train / val / test, PSNR logs, checkpoint rotation; the training
    PSNR improves and a reloaded checkpoint reproduces the network bit for bit."""
    import cdlnet_video_amd as cva
    torch.manual_seed(0)
    net = cva.CDLNet(K=4, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=True).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.9)
    loaders = _loaders((4, 1, 40, 72), 4, 0)
    hist = cva.fit(net, opt, loaders, sched=sched, epochs=3, device=torch.device("cuda"), save_dir=str(tmp_path),
                   clip_grad=5e-2, noise_std=(20, 30), val_freq=1, save_freq=1, verbose=False, log=lambda *_: None)
    train = [v for e, p, v in hist if p == "train"]
    assert len(train) == 3 and all(torch.isfinite(torch.tensor(train)))
    assert train[-1] > train[0], train                         # it learns
    assert [p for e, p, _ in hist if e == 3] == ["train", "val", "test"]
    assert set(os.listdir(tmp_path)) >= {"0.ckpt", "net.ckpt", "train.txt", "val.txt", "test.txt"}
    again = cva.CDLNet(K=4, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=False)
    cva.load_ckpt(os.path.join(str(tmp_path), "net.ckpt"), again)
    again = again.cuda()
    y = loaders["val"][0]
    with torch.no_grad():
        a, _ = net(y, 25.0)
        b, _ = again(y, 25.0)
    assert torch.equal(a, b)
    assert float(net.t.min()) >= 0.0                            # project() ran after every step (train.py:102)


def test_fit_backtracks_a_diverging_video_net(tmp_path):
    """3-D net on the fused generic path: after the epoch-1 checkpoint a filter bank is blown up by 1e3, so epoch 2
    collapses (PSNR far below the best, or nan / inf); fit must reload the checkpoint, scale the learning rate by
    0.8, repeat the epoch with the restored weights and finish."""
    import cdlnet_video_amd as cva
    torch.manual_seed(1)
    net = cva.CDLNetVideo(K=2, M=16, P=[3, 5, 5], s=1, C=1, t0=5e-3, adaptive=True, depth=4, init=True).cuda()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    loaders = _loaders((1, 1, 4, 24, 40), 3, 20)
    logs, fired = [], []

    def epoch_fun(epoch):                                       # runs right after net.ckpt of that epoch is written
        if epoch == 1 and not fired:
            fired.append(1)
            with torch.no_grad():
                net.B[0].weight.mul_(1e3)

    hist = cva.fit(net, opt, loaders, epochs=3, device=torch.device("cuda"), save_dir=str(tmp_path), clip_grad=1,
                   noise_std=25, val_freq=10, save_freq=1, verbose=False, backtrack_thresh=1, epoch_fun=epoch_fun,
                   log=logs.append)
    assert [e for e, p, _ in hist if p == "train"] == [1, 2, 2, 3]          # epoch 2 repeated once
    assert sum("Backtracking" in str(m) for m in logs) == 1
    assert open(tmp_path / "backtrack.txt").read().split() == ["2"]
    assert abs(cva.train.getlr(opt)[0] - 0.8e-3) < 1e-12
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert float(net.B[0].weight.abs().max()) < 10.0              # the blown-up bank was replaced by the checkpoint's


def _f12():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "f12_fit_trajectory.npz"))


def _replay(tag, tmp_path, seed, blow_up, psnr_tol=0.011, **kw):
    """Re-run a trajectory the UNMODIFIED reference `fit` produced on CPU (tools/make_golden_fit.py): same initial
    weights, batches, hyper-parameters and -- through the CPU generator -- the same noise draws, every forward and
    backward in the HIP kernels."""
    import cdlnet_video_amd as cva
    f = _f12()
    net = cva.CDLNet(K=3, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=False)
    net.load_state_dict({k[5:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("init/")})
    net = net.cuda()
    opt = torch.optim.Adam(net.parameters(), lr=kw.pop("lr"))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.9)
    loaders = {ph: [b for b in torch.from_numpy(f["data/" + ph])] for ph in ("train", "val", "test")}
    fired = []

    def epoch_fun(epoch):
        if blow_up and epoch == 1 and not fired:
            fired.append(1)
            with torch.no_grad():
                net.B[0].weight.mul_(1e3)

    torch.manual_seed(seed)
    cva.fit(net, opt, loaders, sched=sched, device=torch.device("cuda"), save_dir=str(tmp_path), verbose=False,
            epoch_fun=epoch_fun, log=lambda *_: None, generator=torch.default_generator, **kw)
    for name in ("train.txt", "val.txt", "test.txt"):
        want = [float(v) for v in str(f[f"{tag}/{name}"]).replace(" ", "").split(",") if v]
        got = [float(v) for v in open(tmp_path / name).read().replace(" ", "").split(",") if v]
        assert len(got) == len(want), (name, got, want)
        assert max(abs(a - b) for a, b in zip(got, want)) <= psnr_tol, (name, got, want)  # logged to 3 decimals
    bt = open(tmp_path / "backtrack.txt").read() if os.path.exists(tmp_path / "backtrack.txt") else ""
    assert bt.split() == str(f[f"{tag}/backtrack.txt"]).split()
    lr = cva.train.getlr(opt)
    assert all(abs(a - b) < 1e-12 for a, b in zip(lr, f[f"{tag}/lr"]))
    worst = {}
    for k, v in net.state_dict().items():
        ref = torch.from_numpy(f[f"{tag}/final/{k}"])
        worst[k] = float((v.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
    print(tag, "final weights, max |diff| / max |ref|:", {k: f"{e:.1e}" for k, e in worst.items()})
    return worst


def test_fit_replays_the_reference_trajectory_with_a_backtrack(tmp_path):
    """Fixture f12 `bt`: Adam + StepLR, 3 epochs with a forced divergence after epoch 1 -- the reference wrote
    backtrack.txt = "2", PSNR logs per phase and lr = 2e-3 * 0.9^3 * 0.8; the product must write the same files."""
    worst = _replay("bt", tmp_path, 123, True, lr=2e-3, epochs=3, clip_grad=5e-2, noise_std=(20, 30), val_freq=1,
                    save_freq=1, backtrack_thresh=1, mcsure=False)
    assert max(worst.values()) <= 2e-3, worst           # final weights after ~12 Adam steps (fp32 CPU reference vs split-bf16 GPU)


def test_fit_replays_the_reference_mcsure_trajectory(tmp_path):
    """Fixture f12 `sure`: the unsupervised MC-SURE objective (train.py:87-93), 2 epochs: same log files (the train
    log is -10 log10 of the SURE loss itself, so it pins the objective's value), same learning rate.
    Its divergence term is a finite difference with h = 1e-3: (net(y + h b) - net(y)) / h multiplies the rounding
    differences between the fp32 CPU reference and the split-bf16 matrix-core path (two-term operand splits: ~1e-6
    forward, ~1e-5 in the gradients) by 1e3; the per-step gradient is gated at 2e-4 with both supports
    prescribed in tests/test_gpu_nets.py.  The PSNR trajectory agrees to ~0.02 dB; the WEIGHTS do not agree tightly --
    in directions where the SURE gradient is below that noise Adam's normalised steps differ in sign (measured: filters
    <= 1e-1 of max |w|, thresholds 3e-1 after 6 steps of 1e-3) -- so they are reported, and only bounded by what 6
    Adam steps can move (parity of the weights themselves is the supervised replay above, <= 8e-4)."""
    worst = _replay("sure", tmp_path, 321, False, psnr_tol=0.05, lr=1e-3, epochs=2, clip_grad=5e-2, noise_std=25,
                    val_freq=1, save_freq=1, backtrack_thresh=1, mcsure=True)
    assert all(e == e and e < 1.0 for e in worst.values()), worst


def test_fit_replays_the_reference_mcsure_trajectory_on_the_fp32_tier(tmp_path):
    """The same replay with every network call on the fp32 VALU kernels (`loop.precision_scope("fp32")`): the finite
    difference now multiplies fp32 rounding only (per-step gradient 6e-6 of autograd with both supports prescribed,
    tests/test_gpu_nets.py).  Measured: the weights end 2-3x closer to the reference's (thresholds 1.5e-1 instead of 3e-1
    of max |t|, filters <= 3e-2 instead of 1e-1) but are still not pinned: what is left is not arithmetic precision but the
    objective -- two fp32 evaluations that differ in the last bit shrink a few code elements to the other side of their
    threshold, and (net(y + h b) - net(y)) / h turns every such flip into a jump of 1e3 x that element.  Any two fp32
    implementations of this loop (this one and the CPU reference included) disagree that way; logs, learning rate and
    PSNR trajectory are gated as above, the weights are reported."""
    from cdlnet_video_amd import loop
    from gpu_util import log
    with loop.precision_scope("fp32"):
        worst = _replay("sure", tmp_path, 321, False, psnr_tol=0.05, lr=1e-3, epochs=2, clip_grad=5e-2, noise_std=25,
                        val_freq=1, save_freq=1, backtrack_thresh=1, mcsure=True)
    log(f"f12 sure replay on the fp32 tier: worst weight deviations {worst}")
    assert all(e == e and e < 1.0 for e in worst.values()), worst
