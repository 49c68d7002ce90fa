"""`python bench.py --gpus N` without an external launcher: the parent must start N fresh rank processes BEFORE
anything can initialise the GPU, hand each its own LOCAL_RANK, pass rank 0's stdout through, and fail when a rank
fails.  Also: the gradient exchange is queued from the end of the reverse sweep (loop.on_backward_end)."""
import importlib.util
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeProc:
    def __init__(self, argv, env, stdout, code):
        self.argv, self.env, self.stdout, self.code, self.pid = argv, env, stdout, code, 4242
        self.terminated = False

    def poll(self):
        return self.code

    def terminate(self):
        self.terminated = True


def test_gpus_n_spawns_n_ranks_before_any_gpu_initialisation(monkeypatch):
    bench = _load_bench()
    assert bench.torch is None                         # importing bench.py does not import torch for it
    started = []

    def fake_popen(argv, env=None, stdout=None, **kw):
        # the launcher is still GPU-free at the moment it starts each child
        assert not torch.cuda.is_initialized()
        assert bench.torch is None
        p = _FakeProc(argv, env, stdout, 0)
        started.append(p)
        return p

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("main() must exit with the launcher's code")
    except SystemExit as e:
        assert e.code == 0
    assert len(started) == 4
    assert sorted(p.env["LOCAL_RANK"] for p in started) == ["0", "1", "2", "3"]
    assert [p.env["RANK"] for p in started] == [p.env["LOCAL_RANK"] for p in started]
    assert {p.env["WORLD_SIZE"] for p in started} == {"4"}
    assert {p.env["MASTER_ADDR"] for p in started} == {"127.0.0.1"}
    assert len({p.env["MASTER_PORT"] for p in started}) == 1
    assert all(p.env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" for p in started)
    for p in started:                                  # each child runs this very file with the same flags
        assert p.argv[0] == sys.executable and os.path.samefile(p.argv[1], os.path.join(ROOT, "bench.py"))
        assert p.argv[2:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert started[0].stdout is None                   # rank 0 inherits stdout: its JSON line is the job's
    assert all(p.stdout is sys.stderr for p in started[1:])
    assert not torch.cuda.is_initialized()


def test_a_failing_rank_fails_the_job_and_stops_the_others(monkeypatch):
    bench = _load_bench()
    started = []

    def fake_popen(argv, env=None, stdout=None, **kw):
        rank = int(env["RANK"])
        p = _FakeProc(argv, env, stdout, 7 if rank == 1 else None)   # rank 1 dies, the others keep running
        orig = p.terminate

        def term():
            orig()
            p.code = -15
        p.terminate = term
        started.append(p)
        return p

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    assert bench.launch_ranks(3, ["--gpus", "3"]) == 7
    assert started[0].terminated and started[2].terminated and not started[1].terminated


def test_under_a_launcher_no_second_spawn(monkeypatch):
    """With WORLD_SIZE set (torch.distributed.run, or this file's own launcher) the process IS a rank."""
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    called = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda *a: called.setdefault("launch", True))
    monkeypatch.setattr(bench, "run_rank", lambda args: called.setdefault("rank", args.gpus))
    bench.main()
    assert called == {"rank": 2}


def test_real_launch_without_gpus_exits_non_zero():
    """End to end on this GPU-less container: both ranks fail at device selection, the parent reports it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        return                                          # on a GPU box this is tests/test_gpu_parallel.py's job
    assert r.returncode != 0
    assert "launcher: started 2 ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_backward_end_hook_runs_once_per_pass_after_the_gradients_exist():
    from cdlnet_video_amd import loop

    class Sweep(torch.autograd.Function):              # stands in for UnrolledISTA.backward's last line
        @staticmethod
        def forward(ctx, w):
            return w * 2

        @staticmethod
        def backward(ctx, g):
            loop._queue_backward_end()
            return g * 2

    w = torch.ones(3, requires_grad=True)
    seen = []
    remove = loop.on_backward_end(lambda: seen.append(None if w.grad is None else w.grad.clone()))
    try:
        (Sweep.apply(w).sum() + Sweep.apply(w).sum()).backward()     # two sweeps in one pass (MC-SURE)
        assert len(seen) == 1 and torch.equal(seen[0], torch.full((3,), 4.0))
        Sweep.apply(w).sum().backward()
        assert len(seen) == 2
    finally:
        remove()
    Sweep.apply(w).sum().backward()
    assert len(seen) == 2
