import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz fixture -> dict of torch tensors / python scalars; 'sd/..', 'grad/..', 'after/..' sub-dicts."""
    raw = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {"sd": {}, "grad": {}, "after": {}, "filt": {}}
    for key in raw.files:
        val = raw[key]
        if val.dtype.kind in "US":
            val = str(val)
        elif val.dtype.kind == "f":
            val = torch.from_numpy(np.array(val, dtype=np.float32))
            if val.dim() == 0:
                val = float(val)
        elif val.dtype.kind in "iu":
            val = [int(v) for v in val.ravel()] if val.ndim else int(val)
        for prefix in ("sd", "grad", "after", "filt"):
            if key.startswith(prefix + "/"):
                out[prefix][key[len(prefix) + 1:]] = val
                break
        else:
            out[key] = val
    return out


@pytest.fixture
def golden():
    return load_golden


def rel_err(a, b):
    """max|a-b| / max|b| (the parity metric of SURVEY.md 8(d))."""
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture
def hip_env(monkeypatch):
    """Setter for the library's CDL_* experiment switches: the library snapshots the environment once, so a
    change is followed by cdl_options_reload(); the snapshot is refreshed again after monkeypatch has undone it."""
    import cdlnet_video_amd as cva

    def setenv(name, value):
        monkeypatch.setenv(name, value)
        cva._lib.reload_options()

    yield setenv
    monkeypatch.undo()
    cva._lib.reload_options()
