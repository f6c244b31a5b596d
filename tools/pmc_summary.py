#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into per-kernel HBM bytes per launch.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel_trace.csv]

FETCH_SIZE / WRITE_SIZE are reported in KB (MI355X_MICROARCH.md, HBM/rocprofv3 section); on gfx950 FETCH_SIZE tallies
128-B requests at 64 B, so it is doubled; WRITE_SIZE is used as is.  The two counters need separate passes (TCC
slots).  With a kernel trace of one of the passes the average duration gives the achieved HBM GB/s."""
import csv
import json
import sys
from collections import defaultdict

# single launches and the level-batched *_multi launches of every family
GROUPS = {"conv_gather": ("conv_gather_kernel", "conv_gather_x3_", "conv3x3_x3_kernel", "conv3x3_x3w_", "stem7_h2_"),
          "conv_wgrad": ("conv_wgrad", "stem_wgrad")}


def group_of(name):
    for g, pat in GROUPS.items():
        pats = pat if isinstance(pat, tuple) else (pat,)
        if any(p in name for p in pats):
            return g
    return None


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        g = group_of(r["Kernel_Name"])
        if g:
            acc[g][0] += float(r["Counter_Value"])
            acc[g][1] += 1
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    dur = defaultdict(lambda: [0.0, 0])
    if len(sys.argv) > 4:
        for r in csv.DictReader(open(sys.argv[4])):
            g = group_of(r["Kernel_Name"])
            if g:
                dur[g][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                dur[g][1] += 1
    res = {}
    for g in GROUPS:
        if f[g][1] == 0 or w[g][1] == 0:
            continue
        fb = f[g][0] / f[g][1] * 1024.0 * 2.0
        wb = w[g][0] / w[g][1] * 1024.0
        res[g] = {"launches_profiled": f[g][1], "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb}
        if dur[g][1]:
            avg = dur[g][0] / dur[g][1]
            res[g]["avg_launch_s_profiled"] = avg
            res[g]["achieved_hbm_gb_per_s"] = (fb + wb) / avg / 1e9
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline` (C3); FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                   "requests at 64 B), WRITE_SIZE (KB) as is; averaged over all launches of the kernel in both steps")
    with open(out, "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
