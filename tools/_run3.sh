mkdir -p gpurun_out/r3; export TMPDIR=/tmp
python -m pytest tests/test_gpu_mixed.py tests/test_gpu_precise.py -x -q -m gpu -s > gpurun_out/r3/t_mixed.log 2>&1; rc=$?
grep -E "parity|passed|failed|Error|error" gpurun_out/r3/t_mixed.log | tail -20
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r3/t_mixed.log; exit $rc; fi
rm -rf gpurun_out/r3/stats_mixed
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/stats_mixed -- python3 bench.py --no-cpu-baseline --no-modes --no-kernel-events --steps 7 --warmup 3 --dtype mixed > gpurun_out/r3/bench_mixed2.json 2> gpurun_out/r3/bench_mixed2.err
python3 tools/kstats_top.py gpurun_out/r3/stats_mixed 10 24 > gpurun_out/r3/mixed_top.txt
cat gpurun_out/r3/mixed_top.txt
s=$(find gpurun_out/r3/stats_mixed -name "*kernel_stats.csv" | head -1); cp $s gpurun_out/r3/mixed_kernel_stats.csv; rm -rf gpurun_out/r3/stats_mixed
python3 -c "
import json; d=json.loads(open('gpurun_out/r3/bench_mixed2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])"
