#!/usr/bin/env python3
"""Print the kernel launch sequence of the LAST bench step from a rocprofv3 --kernel-trace CSV: start offset, duration, stream/queue, short name.
usage: kernel_sequence.py <dir with *_kernel_trace.csv> [pattern to mark]"""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: cut at the largest idle gap in the second half? simpler: print the last N launches
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
short = lambda s: re.sub(r"\(anonymous namespace\)::|void |pmi_igemm_args|unsigned short const\*|, ", "", s)[:70]
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
