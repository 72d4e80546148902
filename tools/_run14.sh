for v in "" _gwd0 "" _gwd0; do
echo "lib$v"; PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so python tools/gemm_probe.py --shapes 2056x4096x1024,2056x1024x4096,2056x3072x1024,2056x1024x1024,8192x4096x1024 --iters 50 2>&1 | grep -v amdgpu
done
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" 2>&1 | tail -2
for v in "" _gwd0 "" _gwd0; do PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so python bench.py --no-cpu-baseline --no-modes --steps 15 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16$v', d['ms_per_step'])"; done
