"""Data-parallel training on the GPU box: two fresh worker processes (gloo, sharing the one GPU) run real
train steps through the HIP kernels; replicas must stay bit-identical and the averaged gradient must equal the
single-process gradient of the global batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from gpu_util import check

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_stay_identical_and_match_the_global_batch_gradient(tmp_path):
    import cdlnet_video_amd as cva
    out = str(tmp_path / "dp.pt")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = torch.load(out, weights_only=True)
    assert res["same"]                                   # parameters bit-identical on both ranks after 3 steps
    assert all(c <= 3 for c in res["copies"]), res["copies"]      # adopt(): a few block copies, not one per parameter
    # single process, global batch: the same first-step gradient (mean of the shard means for equal shards)
    torch.manual_seed(100)
    net = cva.CDLNet(K=3, M=32, P=5, s=1, C=1, t0=5e-3, adaptive=True, init=False)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_ not in ("t", "g"):
                p.mul_(0.05)
            elif n_ == "t":
                p.fill_(5e-3)
    net = net.cuda()
    x_all = cva.utils.synthetic_clip((4, 1, 40, 72), seed=7).cuda()
    noise = (torch.randn(x_all.shape, generator=torch.Generator().manual_seed(8)) * 25 / 255).cuda()
    xhat, _ = net(x_all + noise, 25.0)
    torch.mean((x_all - xhat) ** 2).backward()
    for n_, p in net.named_parameters():
        if p.grad is not None:
            check(f"DP world-2 averaged gradient vs global batch: {n_}", res["first_grads"][n_], p.grad, 2e-5)


def test_bench_gpus_2_runs_two_ranks_and_reports_them():
    """`python bench.py --gpus 2` (no external launcher) starts two fresh ranks; here they share
    the one GPU over a gloo group (CDL_DIST_BACKEND=gloo), on a multi-GPU node the same command runs over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CDL_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "4", "--size", "128", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 8 and out["dist_backend"] == "gloo"
    assert out["grad_syncs_per_step"] == 1.0             # the exchange ran once per step, queued by the reverse sweep
    assert out["value"] > 0 and out["steps"] == 2


def test_net_on_a_non_current_device():
    """a module moved to cuda:1 without torch.cuda.set_device(1) must run on cuda:1's stream with
    cuda:1's launch attributes (ops.py makes the tensors' device current per call; the library keeps its launch
    bookkeeping per device).  Needs two visible GPUs; the one-GPU boxes skip it."""
    import cdlnet_video_amd as cva
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    torch.manual_seed(0)
    net = cva.CDLNet(K=3, M=32, P=7, s=1, C=1, t0=5e-3, adaptive=True, init=False)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_ not in ("t", "g"):
                p.mul_(0.05)
    y = torch.rand(2, 1, 40, 72)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        assert torch.cuda.current_device() == 0
        m = net.to(dev)
        xhat, z = m(y.to(dev), 25.0)
        xhat.square().mean().backward()
        outs.append((xhat.detach().cpu(), z.detach().cpu(), m.A[1].weight.grad.detach().cpu().clone()))
        for p in m.parameters():
            p.grad = None
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):                      # tensors of one call on two devices
        net.to("cuda:1")(y.to("cuda:0"), 25.0)
    # a Gabor net (its bank synthesis takes NESTED tuples of tensors) and a dense-tier block on cuda:1.
    # NOTE: this pool's boxes have one GPU, so everything below the skip has not run on hardware.
    torch.manual_seed(1)
    gd = cva.GDLNet(K=2, M=32, P=7, s=1, C=1, t0=5e-3, order=1, adaptive=True, init=False)
    ref = [t.detach().cpu() for t in gd.to("cuda:0")(y.to("cuda:0"), 25.0)]
    assert torch.cuda.current_device() == 0
    got = [t.detach().cpu() for t in gd.to("cuda:1")(y.to("cuda:1"), 25.0)]
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    xb = torch.randn(1, 32, 3, 12, 40) * 0.5
    w1, w2 = torch.randn(32, 32, 3, 3, 3) / 30, torch.randn(32, 32, 3, 3, 3) / 30
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        gb = cva.ops.residual_geometry(xb, w1)
        outs.append(cva.ops.residual_forward(gb, xb.to(dev), w1.to(dev), w2.to(dev))[1].cpu())
    assert torch.equal(outs[0], outs[1])


def test_device_lookup_searches_nested_sequences():
    import cdlnet_video_amd as cva
    t = torch.zeros(1, device="cuda")
    c = torch.zeros(1)
    assert cva.ops._first_cuda_tensor(([(c, c, c, c), (c, t, c, c)], 7, (1, 2)), {}) is t
    assert cva.ops._first_cuda_tensor((c, [c, (c,)]), {"k": [c]}) is None
    assert cva.ops._first_cuda_tensor((c,), {"k": [[t]]}) is t
