#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kstats_top.py <dir> <steps incl. warm-up> [n]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{tot / steps / 1e6:.2f} ms of kernel time per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
