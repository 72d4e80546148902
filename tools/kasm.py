#!/usr/bin/env python3
"""Extract one kernel from a hipcc -save-temps .s file and print its instruction-class string
(M mfma, v valu, r ds_read, w ds_write, L buffer_load, G global/flat, S scratch, . s_waitcnt, | s_barrier, B branch).
    tools/kasm.py file.s <kernel-name-substring> [--dump out.s]"""
import re, sys
path, sub = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sub in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
out = []
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    c = None
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith("scratch_"): c = "S"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("ds_read") or op.startswith("ds_bpermute"): c = "r"
    elif op.startswith("ds_write"): c = "w"
    elif op.startswith("buffer_load"): c = "L"
    elif op.startswith("buffer_store") or op.startswith("global_store"): c = "W"
    elif op.startswith("global_") or op.startswith("flat_"): c = "G"
    elif op == "s_waitcnt": c = "."
    elif op == "s_barrier": c = "|\n"
    elif op.startswith("s_cbranch"): c = "B"
    elif op == "s_nop": c = "n"
    if c:
        out.append(c)
s = "".join(out)
for row in s.split("\n"):
    for i in range(0, len(row), 200):
        print(row[i:i + 200])
    print("-" * 40)
print("lines", len(body), "mfma", s.count("M"), "valu", s.count("v"), "scratch", s.count("S"), "nop", s.count("n"))
