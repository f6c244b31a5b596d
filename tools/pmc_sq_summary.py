#!/usr/bin/env python3
"""Per-dispatch summary of an SQ counter pass (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT):
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), effective clock =
GRBM_GUI_ACTIVE / 8 / duration.

usage: pmc_sq_summary.py <counter_collection.csv> <kernel_trace.csv> <out.csv> [kernel-name substring ...]"""
import collections
import csv
import sys


def main():
    cc, kt, out = sys.argv[1:4]
    pats = sys.argv[4:] or [""]
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
    d = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        if not any(p in r["Kernel_Name"] for p in pats):
            continue
        d.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"][:90], "grid_threads": r["Grid_Size"]})[r["Counter_Name"]] = \
            float(r["Counter_Value"])
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dispatch", "kernel", "grid_threads", "duration_us"] + cols + ["mfma_busy_frac", "clock_ghz"])
        for k, v in d.items():
            t = trace.get(k)
            if t is None or "GRBM_GUI_ACTIVE" not in v:
                continue
            dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
            gui = v["GRBM_GUI_ACTIVE"]
            w.writerow([k, v["kernel"], v["grid_threads"], f"{dur:.1f}"] + [f"{v.get(c, 0):.0f}" for c in cols] +
                       [f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui / 8 * 1024):.3f}", f"{gui / 8 / dur / 1e3:.2f}"])


if __name__ == "__main__":
    main()
