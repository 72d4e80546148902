python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -k standard -s 2>&1 | grep -E "parity|passed|failed"
b() { python bench.py --no-cpu-baseline --no-modes --steps 10 --dtype mixed 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['avg_ms'])"; }
for v in "" _ilv6 _ilv8 "" _ilv6 _ilv8; do PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so b "mixed$v"; done
python bench.py --no-cpu-baseline --no-modes --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'])"
