"""Where one steady-state train step of a rocprofv3 --kernel-trace CSV spends its time: kernel time by name and the idle
time of the GPU between consecutive kernels (the step waiting for the host).  A step starts at a
normalize_kernel launch more than 5 ms after the previous one; the LAST complete step of the trace is analysed.
usage: python tools/gap_report.py <kernel_trace.csv[.gz]> [min_gap_us]"""
import csv
import gzip
import sys
from collections import defaultdict

path = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
norm = [i for i, r in enumerate(rows) if "normalize_kernel" in r["Kernel_Name"]]
starts = [i for k, i in enumerate(norm) if k == 0 or                      # two crops: two launches a few 100 us apart
          int(rows[i]["Start_Timestamp"]) - int(rows[norm[k - 1]]["Start_Timestamp"]) > 5_000_000]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
win = rows[starts[-2]:starts[-1]]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win) / 1e6
print(f"step: span {span:.2f} ms (first launch to last kernel end), kernel time {busy:.2f} ms, {len(win)} kernels")
by = defaultdict(lambda: [0, 0.0])
for r in win:
    k = short(r["Kernel_Name"])
    by[k][0] += 1
    by[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("kernel time by name:")
for k, (n, t) in sorted(by.items(), key=lambda e: -e[1][1])[:28]:
    print(f"  {t:8.3f} ms {n:5d} x {t / n * 1e3:9.1f} us  {k}")
end = int(win[0]["End_Timestamp"])
gaps = defaultdict(lambda: [0, 0.0])
big, total = [], 0.0
for prev, r in zip(win, win[1:]):
    g = (int(r["Start_Timestamp"]) - end) / 1e3
    end = max(end, int(r["End_Timestamp"]))
    if g <= 0:
        continue
    total += g
    gaps[(short(prev["Kernel_Name"]), short(r["Kernel_Name"]))][0] += 1
    gaps[(short(prev["Kernel_Name"]), short(r["Kernel_Name"]))][1] += g
    if g >= thr:
        big.append((g, short(prev["Kernel_Name"]), short(r["Kernel_Name"])))
print(f"idle between kernels: {total / 1e3:.2f} ms")
for g, a, b in sorted(big, reverse=True)[:15]:
    print(f"  {g:9.1f} us  after {a}  before {b}")
print("idle by (previous, next) kernel:")
for k, (n, g) in sorted(gaps.items(), key=lambda e: -e[1][1])[:12]:
    print(f"  {g / 1e3:8.3f} ms in {n:5d} gaps  {k[0]}  ->  {k[1]}")
