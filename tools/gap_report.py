"""Idle time of the GPU between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream): where the step waits
for the host.  usage: python tools/gap_report.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = defaultdict(lambda: [0, 0.0])
big = []
end = int(rows[0]["End_Timestamp"])
total_gap = 0.0
for prev, r in zip(rows, rows[1:]):
    g = (int(r["Start_Timestamp"]) - end) / 1e3
    end = max(end, int(r["End_Timestamp"]))
    if g <= 0:
        continue
    total_gap += g
    key = (prev["Kernel_Name"][:60], r["Kernel_Name"][:60])
    gaps[key][0] += 1
    gaps[key][1] += g
    if g >= thr:
        big.append((g, prev["Kernel_Name"][:70], r["Kernel_Name"][:70]))
print(f"span {(t1 - t0) / 1e6:.2f} ms, kernel busy {busy / 1e6:.2f} ms, idle {total_gap / 1e3:.2f} ms over {len(rows)} kernels")
print("largest single gaps:")
for g, a, b in sorted(big, reverse=True)[:25]:
    print(f"  {g:9.1f} us  after {a}  before {b}")
print("gap totals by (previous, next) kernel:")
for k, (n, g) in sorted(gaps.items(), key=lambda e: -e[1][1])[:25]:
    print(f"  {g / 1e3:8.3f} ms in {n:5d} gaps  {k[0]}  ->  {k[1]}")
