#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <outdir> <variant> [bench args]  -- same-box A/B of libperceptor_hip.so against libperceptor_hip_<variant>.so:
# two conv probe shapes and two alternating bench pairs
out=$1; v=$2; shift 2; mkdir -p $out
N=$PWD/perceptor_amd/csrc/libperceptor_hip.so; O=$PWD/perceptor_amd/csrc/libperceptor_hip_$v.so
for shape in "--hw 256 --cin 256 --cout 256" "--hw 512 --cin 128 --cout 128"; do
  for L in $N $O; do r=$(PMI_LIB=$L python tools/conv_probe.py $shape --pro 1 --res 1 --stats 1 --rounds 2 --iters 30 2>&1 | tail -n 1) || exit 1; echo "$(basename $L) $r" >> $out/ab.txt; done
done
for i in 1 2; do for L in $N $O; do
  PMI_LIB=$L python bench.py --no-cpu-baseline --no-modes "$@" 2>>$out/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $L)', d['ms_per_step'], d['roofline']['frac'])" >> $out/ab.txt || exit 1
done; done
cat $out/ab.txt
