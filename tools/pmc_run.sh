#!/bin/bash
# usage: tools/pmc_run.sh <outdir> "<counters pass 1>" ["<counters pass 2>" ...] -- python3 prog args...   (GPU box; one rocprofv3 run per pass)
out=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
export TMPDIR=/tmp
i=0
for p in "${passes[@]}"; do
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $out/pass$i -- "$@" > $out.pass$i.log 2>&1 || { tail -5 $out.pass$i.log; exit 1; }
  i=$((i+1))
done
