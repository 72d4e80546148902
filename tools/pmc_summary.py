#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one pass each) into profiles/rNN_pmc_hbm.json.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/r01_pmc_hbm.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so reads = 2 * FETCH_SIZE; Infinity-Cache hits
are counted, so the figure is fabric traffic (an upper bound on HBM traffic).
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").replace("(anonymous namespace)::", "")
            agg[name].append(float(r["Counter_Value"]))
    return agg


def kernel_src_sha16():
    """Identity of the kernels the HBM figures belong to: bench.py reports `roofline.traffic` from this file only while the conv sources are unchanged."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "perceptor_amd", "csrc")
    h = hashlib.sha256()
    for n in ("conv_wd.hip", "conv3x3.hip", "common.h"):
        with open(os.path.join(root, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    fetch, write, out = sys.argv[1:4]
    fa, wa = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {"_note": "per-launch means over one `bench.py --steps 2 --warmup 1` run per counter; KiB as reported; "
                    "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction)",
           "_kernel_src_sha16": kernel_src_sha16()}
    res["_bench_args"] = sys.argv[4] if len(sys.argv) > 4 else ""
    for k in sorted(set(fa) | set(wa), key=lambda k: -(2 * sum(fa.get(k, [0])) + sum(wa.get(k, [0])))):
        f, w = fa.get(k, []), wa.get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        res[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_mean": round(fm, 1), "WRITE_SIZE_KiB_mean": round(wm, 1),
                  "hbm_bytes_per_launch": round((2 * fm + wm) * 1024)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in list(res.items())[:8]}, indent=1))


if __name__ == "__main__":
    main()
