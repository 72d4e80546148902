export PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip_stamps.so
python tools/gemm_probe.py --shapes 2056x4096x1024,2056x1024x4096,2056x3072x1024,2056x1024x1024,8192x4096x1024 --stamps 1 --iters 30 2>&1 | grep -v amdgpu
