#!/bin/bash
# Timing decomposition of conv3x3_wd_kernel's main loop by removal: diagnostic libraries built with -DWD_DIAG=<mask> (1 no weight loads
# in the loop, 2 no fragment reads, 4 no patch staging, 8 no chunk barrier; results are wrong, timing is the point), one probe shape each.
# Build: python -c "from perceptor_amd.csrc.build import build; [build(variant=f'd{k}', extra_flags=[f'-DWD_DIAG={k}']) for k in (1,2,4,8,5,7,15)]"
out=${1:-gpurun_out/decomp.txt}; mkdir -p $(dirname $out); : > $out
for shape in "--hw 256 --cin 256 --cout 256" "--hw 512 --cin 128 --cout 128"; do
  for pro in 1 0; do
    for v in "" _d1 _d2 _d4 _d16 _d8 _d5 _d7 _d15; do
      L=$PWD/perceptor_amd/csrc/libperceptor_hip$v.so; [ -f $L ] || continue
      r=$(PMI_LIB=$L python tools/conv_probe.py $shape --pro $pro --res 1 --stats 1 --rounds 2 --iters 30 2>&1 | tail -n 1) || exit 1
      echo "diag${v:-_0} $r" >> $out
    done
  done
done
cat $out
