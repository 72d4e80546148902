python -m pytest tests/test_gpu_mixed.py tests/test_gpu_kernels.py tests/test_gpu_adm.py -x -q -m gpu 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline --no-modes --steps 10 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['roofline']['avg_ms'])"; }
b --dtype mixed
b --dtype bf16
b --dtype f16
b --dtype mixed
b --dtype bf16
b --dtype f16
./tools/mfma_peak | tail -8
