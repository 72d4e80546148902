#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats run: tools/kstats.py <dir> <steps incl. warmup>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 36]:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print("%8.3f ms/step %7.1f calls/step %9.1f us  %s" % (float(r["TotalDurationNs"]) / n / 1e6, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, name[:120]))
print("total ms/step %.3f, launches/step %.0f" % (tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
