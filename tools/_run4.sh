mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -k "standard" -s 2>&1 | grep -E "parity|passed|failed" 
for v in "" _ilv3 _ilv4; do
  for i in 1 2; do
  PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so python bench.py --no-cpu-baseline --no-modes --steps 10 --dtype mixed 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed$v', d['ms_per_step'], d['roofline']['avg_ms'])"
  done
done
for v in "" _ilv3 _ilv4; do
  PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so python bench.py --no-cpu-baseline --no-modes --steps 15 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16$v', d['ms_per_step'], d['roofline']['avg_ms'])"
done
