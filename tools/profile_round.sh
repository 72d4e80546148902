#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag> [bench args...]      e.g. r03   /   r03_mixed --dtype mixed   -> gpurun_out/<tag>/...
# 1. the bench line (default run: with cpu_baseline and modes); 2. rocprofv3 --kernel-trace --stats of the same command; 3. PMC passes (FETCH_SIZE,
# WRITE_SIZE, MfmaUtil, LdsUtil), each its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md: the TCC counters do not fit one pass).
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
tail -c 900 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-modes "$@" > $out/bench_under_rocprof.json 2> $out/stats.err
for c in FETCH_SIZE WRITE_SIZE MfmaUtil LdsUtil; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --no-kernel-events "$@" > $out/pmc_$c.json 2> $out/pmc_$c.err
  echo "pmc $c done"
done
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $w $out/pmc_hbm.json "$*" > /dev/null
python3 tools/pmc_kernel.py $out/pmc_MfmaUtil > $out/pmc_mfma_util.txt
python3 tools/pmc_kernel.py $out/pmc_LdsUtil > $out/pmc_lds_util.txt
s=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $s $out/kernel_stats.csv
# the raw traces are large: keep only the summaries
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_MfmaUtil $out/pmc_LdsUtil
ls -la $out
