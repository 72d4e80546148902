#!/bin/bash
# usage (GPU box): tools/other_configs.sh <outdir>   -- the bench line and a rocprofv3 --kernel-trace --stats summary for BASELINE configs c1..c4,
# and the three UNet forward + input-gradient lines (bench.py --backward); copy the results to profiles/ with the names profiles/README.md lists
set -e
out=${1:-gpurun_out/others}; mkdir -p $out; export TMPDIR=/tmp
declare -A ARGS=( [c1]="--config c1 --graph" [c2]="--config c2 --graph" [c3]="--config c3" [c4]="--config c4" )
for c in c1 c2 c3 c4; do
  python3 bench.py ${ARGS[$c]} > $out/bench_$c.json 2> $out/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py ${ARGS[$c]} --no-cpu-baseline --no-modes > $out/bench_${c}_under_rocprof.json 2> $out/stats_$c.err
  cp $(find $out/stats_$c -name "*kernel_stats.csv" | head -1) $out/bench_${c}_kernel_stats.csv; rm -rf $out/stats_$c
  echo "$c done"
done
python3 bench.py --config c3 --backward --no-cpu-baseline > $out/bench_c3_backward.json 2> $out/bw.err
python3 bench.py --config c5-noclip --backward --no-cpu-baseline > $out/bench_c5_unet_backward.json 2>> $out/bw.err
python3 bench.py --config c2 --backward --no-cpu-baseline > $out/bench_c2_backward.json 2>> $out/bw.err
for f in $out/bench_c?.json $out/bench_*backward.json; do python3 -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d.get('tape_gb'))"; done
