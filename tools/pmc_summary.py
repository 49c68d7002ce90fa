#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) over
tools/bench_kernels.py into profiles/<tag>_pmc_traffic.json: HBM bytes per launch of every fused kernel.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <layout>

The passes run ONE code layout (BK_LAYOUTS=<layout>); the output records it and the sha1 of cdl_fused2d.hip, so
that bench.py only quotes a profile of the kernel revision and layout it is timing.

Counter values are in KiB.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half
of the bytes of coalesced streaming reads); the same pass calibrates that on __amd_rocclr_copyBuffer.
"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)

MODES = {0: "FWD", 1: "FIRST", 2: "BWD"}
PRECS = {0: "split3", 1: "bf16"}


LAYS = {0: "nchw", 1: "blocked", 2: "blocked_bf16"}


def label(name):
    m = re.search(r"k_stage<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>", name)   # <MT, PREC, MODE, LIN, LOUT, DA>
    if m:                                                    # (the reverse stage is profiled in its shipped form: with dA_k)
        return f"k_stage<{MODES[int(m.group(3))]},{PRECS[int(m.group(2))]}>"
    m = re.search(r"k_wgrad2d<(\d+), (\d+), (\d+)>", name)                      # <MT, PREC, LAY>
    if m:
        return f"k_wgrad2d<{PRECS[int(m.group(2))]}>"
    for k in ("k_assemble", "k_support_map", "__amd_rocclr_copyBuffer"):
        if k in name:
            return k
    return None


def fold(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        lab = label(row["Kernel_Name"])
        if lab:
            acc[lab].append(float(row["Counter_Value"]) * 1024.0)
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}          # median launch


def kernel_source_sha():
    import hashlib
    import os
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cdlnet-video_amd", "csrc",
                       "cdl_fused2d.hip")
    return hashlib.sha1(open(src, "rb").read()).hexdigest()[:16]


def main():
    fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"fetch_reported": int(f), "fetch_corrected": int(2 * f), "write": int(w),
                      "traffic": int(2 * f + w)}
    out = {"_note": __doc__.strip().split("\n\n")[-1].replace("\n", " ") +
           " Median launch per kernel; k_stage<FWD,split3> mixes launches with and without the support map"
           " (67 MB more written with it).",
           "shape": {"N": 64, "M": 64, "H": 256, "W": 256, "P": 7}, "layout": sys.argv[4],
           "kernel_source_sha": kernel_source_sha(), "kernels": kernels}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
