#!/usr/bin/env python3
"""Mean of each PMC counter per kernel from rocprofv3 counter_collection.csv files (one or more passes).
    python tools/pmc_kernel.py <dir> [kernel-name substring]"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in agg.items():
    if filt not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
