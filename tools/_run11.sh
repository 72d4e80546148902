mkdir -p gpurun_out/r3
bash tools/sq_decomp.sh gpurun_out/r3/sq 2>&1 | tail -25
python bench.py --no-cpu-baseline --no-modes --steps 8 --dtype precise 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('precise', d['ms_per_step'])"
