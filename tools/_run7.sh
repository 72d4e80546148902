python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -s 2>&1 | grep -E "parity|passed|failed|Error" 
b() { python bench.py --no-cpu-baseline --no-modes --steps 10 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['roofline']['avg_ms'])"; }
b --dtype mixed
b --dtype mixed --opt 14=0
b --dtype mixed
b --dtype mixed --opt 14=0
export PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip_stamps.so
for args in "--mixed single --hw 256 --cin 256 --cout 256 --res 1" "--mixed single --hw 256 --cin 256 --cout 256 --res 0" "--mixed dbl --hw 512 --cin 128 --cout 128 --res 1"; do
  echo "== $args"
  python tools/conv_probe.py $args --stamps 1 --rounds 1 --iters 10 2>&1 | grep -E "mixed|prologue|mainloop|epilogue"
done
