set -e
mkdir -p gpurun_out/sd
python -m pytest tests/test_gpu_sd.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/sd/bench_c4.json 2> gpurun_out/sd/bench_c4.err || { tail -20 gpurun_out/sd/bench_c4.err; exit 1; }
cat gpurun_out/sd/bench_c4.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sd/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/sd/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py gpurun_out/sd/prof2 12 30
