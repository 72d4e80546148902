#!/bin/bash
# usage (GPU box): tools/sq_decomp.sh <out dir>   -- SQ counter decomposition of the weights-direct conv (256 -> 256 @ 256x256 x 8, f16) with and
# without the fused GroupNorm-apply + SiLU prologue.  One rocprofv3 --pmc pass per counter group, the program directly after `--`.
set -e
out=$1; mkdir -p $out; export TMPDIR=/tmp
for pro in 0 1; do
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
    d=$out/pmc_pro${pro}_$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 tools/conv_probe.py --dtype f16 --pro $pro --res 1 --stats 1 --hw 256 --cin 256 --cout 256 --rounds 1 --iters 5 > $d.log 2>&1 || { echo "group failed: $grp"; tail -3 $d.log; }
  done
done
python3 - <<'PY' $out
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for pro in (0, 1):
    for path in glob.glob(f"{out}/pmc_pro{pro}_*/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "conv3x3_wd_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            tab[k][pro] = sum(v) / len(v)
with open(f"{out}/sq_decomposition.txt", "w") as f:
    f.write("conv3x3_wd_kernel<F16, PRO, true, 8, 64>, 256 -> 256 @ 256x256 x 8, residual + statistics; mean per launch over 6 launches\n")
    f.write(f"{'counter':32s} {'pro 0 (no prologue)':>22s} {'pro 1 (GN + SiLU)':>22s} {'ratio':>8s}\n")
    for k in sorted(tab):
        a, b = tab[k].get(0), tab[k].get(1)
        f.write(f"{k:32s} {a if a is not None else float('nan'):22.4g} {b if b is not None else float('nan'):22.4g} {(b / a) if a and b else float('nan'):8.3f}\n")
print(open(f"{out}/sq_decomposition.txt").read())
PY
rm -rf $out/pmc_pro*
