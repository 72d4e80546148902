mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -s > gpurun_out/r3/t_mixed.log 2>&1; rc=$?
tail -40 gpurun_out/r3/t_mixed.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-modes --steps 8 --dtype mixed --dump-kernels gpurun_out/r3/mixed_kernels.txt > gpurun_out/r3/bench_mixed.json 2> gpurun_out/r3/bench_mixed.err; echo rc=$?
tail -c 1500 gpurun_out/r3/bench_mixed.json; tail -5 gpurun_out/r3/bench_mixed.err
