"""Imported FIRST by the timing-ablation tools: points the package at the probe build of the library
(`make -C cdlnet-video_amd/csrc libcdlnet_hip_ablate.so`, -DCDL_ABLATE), the only build in which the CDL_FUSED_DEBUG /
CDL_DENSE_DEBUG switches exist.  The product library ignores those variables."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABLATE_LIB = os.path.join(ROOT, "cdlnet-video_amd", "csrc", "libcdlnet_hip_ablate.so")
if not os.path.exists(ABLATE_LIB):
    sys.exit(f"{ABLATE_LIB} missing: run `make -j8 -C cdlnet-video_amd/csrc libcdlnet_hip_ablate.so` first")
os.environ["CDLNET_HIP_LIB"] = ABLATE_LIB
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
