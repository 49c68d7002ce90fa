#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch use of one HIP source, as the compiler reports it (no GPU needed):

    python tools/kernel_resources.py cdl_fusedg.hip [extra hipcc flags] [--grep PATTERN]

prints: kernel, VGPRs, AGPRs, SGPRs, scratch bytes per lane, static LDS bytes, occupancy (waves per SIMD)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cdlnet-video_amd", "csrc")
args = sys.argv[1:]
pat = None
if "--grep" in args:
    i = args.index("--grep")
    pat = re.compile(args[i + 1])
    del args[i:i + 2]
src, extra = args[0], args[1:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", *extra,
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src]
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
keys = (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"))
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if " error: " in line:
        print(line)
    for key, p in keys:
        m = re.search(p, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
for r, name in zip(rows, names):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    if pat and not pat.search(name):
        continue
    print(f"{name:64s} vgpr {r.get('vgpr', -1):3d} agpr {r.get('agpr', -1):3d} sgpr {r.get('sgpr', -1):3d} "
          f"scratch {r.get('scratch', -1):4d} lds {r.get('lds', -1):6d} occ {r.get('occ', -1)}")
