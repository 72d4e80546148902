#!/bin/bash
# usage: tools/copy_profiles.sh <gpurun_out tag> <profiles prefix> [suffix]   e.g.  r03 r03 ""   /   r03_mixed r03 _mixed
# copies the summaries of tools/profile_round.sh into profiles/ under the names bench.py and the README expect
t=$1; p=$2; s=$3
cp gpurun_out/$t/bench.json profiles/${p}_bench_c5$s.json
cp gpurun_out/$t/kernel_stats.csv profiles/${p}_bench_c5${s}_kernel_stats.csv
cp gpurun_out/$t/bench_under_rocprof.json profiles/${p}_bench_c5${s}_under_rocprof.json
cp gpurun_out/$t/pmc_hbm.json profiles/${p}_pmc_hbm$s.json
cp gpurun_out/$t/pmc_mfma_util.txt profiles/${p}_pmc_mfma_util$s.txt
cp gpurun_out/$t/pmc_lds_util.txt profiles/${p}_pmc_lds_util$s.txt
