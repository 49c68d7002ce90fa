#!/usr/bin/env python3
"""Fold rocprofv3 outputs of ONE workload into a compact per-kernel table (JSON on stdout):

    python tools/pmc_fold.py <tag> <kernel_stats.csv> [<counter_collection.csv> ...]

kernel_stats.csv comes from `--kernel-trace --stats`, each counter_collection.csv from its own `--pmc` pass
(counters are never collected together with the trace, as the pool's rules require).  Per kernel: calls,
average duration, and the MEDIAN per-launch value of every counter found.  FETCH_SIZE / WRITE_SIZE are KiB;
`hbm_bytes` = 2 * FETCH_SIZE + WRITE_SIZE in bytes (MI355X_MICROARCH.md, HBM: gfx950 reports half of the
bytes of a coalesced streaming read).  Kernel names are shortened to `name<template args>`.
"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_][\w:]*)(<[^(]*>)?\(", name)
    if not m:
        return name[:60]
    base = m.group(1).split("::")[-1]
    return base + (m.group(2) or "")


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    table = collections.OrderedDict()
    for row in csv.DictReader(open(stats)):
        k = short(row["Name"])
        e = table.setdefault(k, {"calls": 0, "total_ns": 0})
        e["calls"] += int(row["Calls"])
        e["total_ns"] += int(row["TotalDurationNs"])
    for e in table.values():
        e["avg_us"] = round(e["total_ns"] / e["calls"] / 1e3, 2)
    total = sum(e["total_ns"] for e in table.values())
    for e in table.values():
        e["share"] = round(e.pop("total_ns") / total, 4)
    for path in sys.argv[3:]:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, counters in acc.items():
            if k in table:
                for cn, vals in counters.items():
                    table[k][cn] = sorted(vals)[len(vals) // 2]
    for e in table.values():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
            e["hbm_GBps"] = round(e["hbm_bytes"] / e["avg_us"] / 1e3, 1)
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts over the 4 SIMDs of a CU, SQ_BUSY_CU_CYCLES per CU: / 4 = the share of a
            # SIMD's cycles in which its matrix pipe is busy (the un-normalised ratio reads > 1)
            e["mfma_busy_frac_per_simd"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CU_CYCLES"] / 4.0, 4)
    keep = {k: v for k, v in table.items() if v["share"] >= 0.002}
    print(json.dumps({"tag": tag, "kernels": keep}, indent=1))


if __name__ == "__main__":
    main()
