export PMI_LIB=$PWD/perceptor_amd/csrc/libperceptor_hip_stamps.so
for args in "--mixed dbl --hw 512 --cin 128 --cout 128 --res 1" "--mixed single --hw 512 --cin 128 --cout 128 --res 1" "--mixed single --hw 256 --cin 256 --cout 256 --res 1" "--mixed dbl --hw 256 --cin 256 --cout 256 --res 1" "--dtype f16 --pro 1 --res 1 --stats 1 --hw 256 --cin 256 --cout 256" "--dtype f16 --pro 1 --res 1 --stats 1 --hw 512 --cin 128 --cout 128"; do
  echo "== $args"
  python tools/conv_probe.py $args --stamps 1 --rounds 1 --iters 10 2>&1 | grep -v amdgpu.ids
done
