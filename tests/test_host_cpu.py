"""CPU-only (no GPU) checks: host logic, checkpoint surface, the C-ABI library and its header."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, rel_err
import cdlnet_video_amd as cva


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cdlnet_hip.h")).read()
    declared = set(re.findall(r"\b(cdl_[a-z0-9_]+)\s*\(", header)) - {"cdl_geom"}
    assert len(declared) >= 13
    lib = ctypes.CDLL(cva._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/cdlnet_hip.h but not exported"
    bound = set(cva._lib.SIGNATURES) | set(cva._lib.SIZE_T_FUNCS) | {"cdl_version"}
    assert declared <= bound, declared - bound          # every entry point has a Python binding
    lib.cdl_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cdl_version()
    assert ctypes.sizeof(cva._lib.Geom) == 15 * 4


def test_exact_fp32_switch_is_per_thread_and_scoped():
    """cdl_set_exact_fp32 (host-only state, no GPU needed): returns the previous setting, is private to the calling thread, and
    the Python scopes restore what they found -- also when the body raises."""
    import threading
    lib = cva._lib.lib()
    assert lib.cdl_set_exact_fp32(0) == 0
    with cva.ops.exact_fp32():
        seen = []
        t = threading.Thread(target=lambda: seen.append(lib.cdl_set_exact_fp32(0)))
        t.start()
        t.join()
        assert seen == [0]                                   # another thread still sees the default
        assert lib.cdl_set_exact_fp32(1) == 1                # this one sees the scope
        with cva.ops.exact_fp32(False):                      # a no-op block
            assert lib.cdl_set_exact_fp32(1) == 1
    assert lib.cdl_set_exact_fp32(0) == 0
    with pytest.raises(ZeroDivisionError):
        with cva.ops.exact_fp32():
            1 / 0
    assert lib.cdl_set_exact_fp32(0) == 0
    assert "fp32" in cva.loop.ARITHMETIC
    with cva.loop.precision_scope("fp32"):
        assert cva.loop.PRECISION == "fp32"
    assert cva.loop.PRECISION == "split3"
    with pytest.raises(ValueError):
        cva.loop.precision_scope("fp64")


def test_no_cpu_compute_path():
    torch.manual_seed(0)
    net = cva.CDLNet(K=2, M=4, P=5, init=False)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.rand(1, 1, 16, 16))
    with pytest.raises(RuntimeError):
        net.project()
    with pytest.raises(RuntimeError):
        cva.ops.preprocess(torch.rand(1, 1, 8, 8), 1)
    vid = cva.CDLNetVideo(K=2, M=3, P=3, init=False)
    with pytest.raises(RuntimeError):
        vid(torch.rand(1, 1, 4, 8, 8))
    gd = cva.GDLNet(K=2, M=3, P=5, init=False)
    with pytest.raises(RuntimeError):
        gd(torch.rand(1, 1, 8, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cdlnet-video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


def test_seeded_init_equals_reference():
    """torch.manual_seed(s); Cls(...) reproduces the reference's weights (fixtures from the reference)."""
    g = load_golden("f8_init_2d")
    K, M, P, s, C = g["hyper"]
    torch.manual_seed(g["seed"])
    net = cva.CDLNet(K=K, M=M, P=P, s=s, C=C, t0=g["t0"], adaptive=True, init=True)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())
    for k, v in g["sd"].items():
        assert sd[k].shape == v.shape
        assert rel_err(sd[k], v) < 1e-6, k
    g = load_golden("f8_init_3d")
    K, M, _, s, C = g["hyper"]
    torch.manual_seed(g["seed"])
    net = cva.CDLNetVideo(K=K, M=M, P=g["P3"], s=s, C=C, t0=g["t0"], adaptive=True, depth=g["depth"])
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())
    for k, v in g["sd"].items():
        assert rel_err(sd[k], v) < 1e-6, k


def test_state_dict_surface():
    net = cva.CDLNet(K=3, M=5, P=7, s=2, C=3, init=False)
    sd = net.state_dict()
    assert set(sd) == {"t", "g", "D.weight"} | {f"{b}.{k}.weight" for b in "AB" for k in range(3)}
    assert sd["A.1.weight"].shape == (5, 3, 7, 7) and sd["B.1.weight"].shape == (5, 3, 7, 7)
    assert sd["t"].shape == (3, 2, 5, 1, 1)
    assert net.D is net.B[0]
    assert len(list(net.parameters())) == 2 + 6                  # D de-duplicated
    # upstream checkpoints without `g` load
    legacy = {k: v for k, v in sd.items() if k != "g"}
    net.load_state_dict(legacy)
    vid = cva.CDLNetVideo(K=2, M=4, P=[9, 9, 5], s=2, init=False)
    sdv = vid.state_dict()
    assert "g" not in sdv and sdv["t"].shape == (2, 2, 4, 1, 1, 1)
    assert sdv["A.0.weight"].shape == (4, 1, 9, 9, 5)
    assert cva.CDLNetVideo(K=1, M=2, P=7, init=False).P == (7, 7, 7)     # args3d.json's int P
    res = cva.CDLNetVideo(K=2, M=4, P=3, init=False, residual=True)        # net.py:146-151
    assert res.state_dict()["residual_blocks.1.conv2.weight"].shape == (4, 4, 3, 3, 3)
    assert isinstance(res.residual_blocks[0], cva.ResidualBlock) and isinstance(res.residual_blocks[0].relu, torch.nn.ReLU)
    with pytest.raises(RuntimeError):                                      # no CPU compute path
        res(torch.zeros(1, 1, 3, 8, 8))
    with pytest.raises(ValueError):
        cva.ResidualBlock(4, 8)
    for attr in ("K", "M", "P", "s", "t0", "adaptive"):
        assert hasattr(net, attr) and hasattr(vid, attr)


def test_gabor_sharing_and_state_dict():
    g = load_golden("f5_gabor_shared")
    K, M, P, s, C = g["hyper"]
    net = cva.GDLNet(K=K, M=M, P=P, s=s, C=C, order=g["order"], shared=g["shared"], init=False)
    assert list(net.state_dict().keys()) == list(g["sd"].keys())
    assert net.A[2].alpha is net.A[0].alpha and net.B[2].alpha is net.B[1].alpha
    assert net.B[1].alpha is not net.B[0].alpha
    assert net.A[1].a is net.A[0].a and net.B[2].psi is net.B[0].psi
    names = [n for n, _ in net.named_parameters()]
    assert names == list(g["grad"].keys())
    net.load_state_dict(g["sd"])
    plain = cva.GDLNet(K=2, M=3, P=5, order=1, shared="", init=False)
    assert plain.A[1].a is not plain.A[0].a


def test_gabor_seeded_init_runs_power_method():
    torch.manual_seed(3)
    net = cva.GDLNet(K=2, M=4, P=5, s=1, C=1, order=2, shared="a_psi_w0_alpha", init=True)
    assert torch.isfinite(net.A[0].alpha).all()


def test_build_model_from_args_json_shapes():
    shipped = {"type": "CDLNet", "model": {"adaptive": True, "K": 20, "M": 32, "C": 1, "P": 7, "s": 1},
               "paths": {"save": "x", "ckpt": "x/net.ckpt"}, "train": {"opt": {"lr": 1e-3},
                                                                        "sched": {"gamma": 0.95, "step_size": 50}}}
    net = cva.build_model(shipped)               # ckpt path given -> init=False, like train.py:185
    assert (net.K, net.M, net.P, net.s) == (20, 32, 7, 1)
    vid = cva.build_model({"type": "CDLNetVideo", "paths": {"ckpt": None},
                           "model": {"adaptive": True, "K": 2, "M": 4, "C": 1, "P": [9, 9, 5], "s": 2,
                                     "t0": 0, "depth": 16, "init": False, "residual": False}})
    assert vid.P == (9, 9, 5)
    jdd = cva.build_model({"type": "JDD_CDLNet", "paths": {"ckpt": "c"},
                           "model": {"adaptive": True, "K": 2, "M": 4, "C": 3, "P": 7, "s": 1}})
    assert isinstance(jdd, cva.CDLNet)
    with pytest.raises(NotImplementedError):
        cva.build_model({"type": "DnCNN", "model": {}})


def test_checkpoint_round_trip(tmp_path):
    net = cva.CDLNet(K=2, M=3, P=5, init=False)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.9)
    path = str(tmp_path / "net.ckpt")
    cva.save_ckpt(path, net, 7, opt, sched)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"epoch", "net_state_dict", "opt_state_dict", "sched_state_dict"}
    other = cva.CDLNet(K=2, M=3, P=5, init=False)
    _, _, _, epoch = cva.load_ckpt(path, other)
    assert epoch == 7
    assert torch.equal(other.A[1].weight, net.A[1].weight)


def test_data_helpers_match_reference(golden):
    g = golden("f9_helpers")
    assert torch.equal(cva.gen_bayer_mask(torch.zeros(1, 3, 6, 8)), g["bayer"])
    assert abs(float(cva.gen_bayer_mask(torch.zeros(2, 3, 8, 8)).mean()) - 1 / 3) < 1e-6
    for i in range(5):
        H, W, s = g["pads2_in"][3 * i:3 * i + 3]
        assert list(cva.ops.stride_pads((H, W), s)) == g["pads2"][4 * i:4 * i + 4]
    for i in range(3):
        D, H, W, s = g["pads3_in"][4 * i:4 * i + 4]
        assert list(cva.ops.stride_pads((D, H, W), s)) == g["pads3"][6 * i:6 * i + 6]
    x = torch.rand(4, 1, 8, 8)
    y, sig = cva.awgn(x, (20, 30), torch.Generator().manual_seed(0))
    assert sig.shape == (4, 1, 1, 1) and float(sig.min()) >= 20 and float(sig.max()) <= 30
    y, sig = cva.awgn3d(torch.rand(2, 1, 4, 8, 8), [5, 6])
    assert sig.shape == (2, 1, 1, 1, 1)
    y, sig = cva.awgn(x, 25)
    assert sig == 25 and y.shape == x.shape
    assert abs(cva.psnr(x, x + 0.1) - 20.0) < 1e-4


def test_csr_modules_surface():
    """CDLNet_CSR / CDLNet_CSRf2 (reference model/net.py:363-568): state_dict keys, parameter names and
    order as in the fixtures generated from the reference; no CPU compute path either."""
    for name, cls in (("c1_csr_chain", cva.CDLNet_CSR), ("c2_csrf2_chain", cva.CDLNet_CSRf2)):
        g = load_golden(name)
        K, M, P, s, C = g["hyper"]
        net = cls(K=K, M=M, P=P, s=s, C=C, t0=0.0, adaptive=True, init=False)
        assert list(net.state_dict().keys()) == list(g["sd"].keys())
        names = [n for n, _ in net.named_parameters()]
        assert [n for n in names if n in g["grad"]] == list(g["grad"].keys())
        assert set(names) - set(g["grad"]) <= {"B2.0.weight"}      # never applied: D is B[0] (net.py:383)
        net.load_state_dict(g["sd"])
        assert net.D is net.B[0]
        with pytest.raises(RuntimeError, match="no CPU"):
            net(torch.rand(1, 1, 16, 16))
    net = cva.build_model({"type": "CDLNet_CSRf2", "paths": {"ckpt": "c"},      # argscsr.json's shape
                           "model": {"adaptive": True, "K": 2, "M": 5, "C": 1, "P": 9, "s": 2, "t0": 0,
                                     "init": True}})
    assert isinstance(net, cva.CDLNet_CSRf2) and net.g1.shape == (2, 2, 5, 1, 1)


def test_awgn_generator_reproduces_the_reference_noise_stream():
    """utils.py:29-41 draws `rand(N,1,1,1)` for sigma and then `randn_like(input)` from the global generator; with
    `generator=torch.default_generator` the product draws the same numbers in the same order (what lets a GPU run
    replay a CPU reference run, tests/test_gpu_trainer.py)."""
    x = torch.rand(3, 1, 6, 5)
    torch.manual_seed(11)
    sigma = 20 + (30 - 20) * torch.rand(len(x), 1, 1, 1)
    want = x + torch.randn_like(x) * (sigma / 255)
    torch.manual_seed(11)
    got, s = cva.utils.awgn(x, (20, 30), torch.default_generator)
    assert torch.equal(got, want) and torch.equal(s, sigma)
    torch.manual_seed(11)
    got2, s2 = cva.utils.awgn(x, (20, 30))                     # no generator: the input's device default, same stream on CPU
    assert torch.equal(got2, want) and torch.equal(s2, sigma)
