"""A short seeded run of tools/fuzz_parity.py inside the -m gpu suite: random geometries (ragged sizes, 2-D / 3-D,
stride 1 / 2, odd channel counts) through the matrix-core tier -- analysis, synthesis, filter gradients incl. the paired
launch, the reverse analysis step, the fused generic stage -- against the fp32 VALU tier of the same library, which
tests/test_gpu_ops.py pins to the oracle."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_geometries_matrix_core_tier_vs_fp32_tier(hip_env):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    try:
        worst, fails = fuzz.run(cases=10, seed=11, verbose=False)
    finally:
        fuzz.setenv("1")
        for k in ("CDL_MFMA_ANALYSIS", "CDL_MFMA_SYNTHESIS", "CDL_MFMA_WGRAD"):
            os.environ.pop(k, None)
        fuzz.cva._lib.reload_options()
    assert not fails, fails
    assert max(v for k, v in worst.items() if k != "fusedg_flips") < 3e-5, worst
