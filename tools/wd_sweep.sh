#!/bin/bash
# A/B of the weights-direct conv3x3 kernel (configs 4, 5) against the round-1 halo kernel (configs 0, 2) on the UNet's main shapes
for shape in "256 256 256" "512 128 128" "512 256 256" "128 512 512" "64 512 512" "256 512 256" "512 384 128"; do
  set -- $shape
  for cfg in ${CFGS:-0 2 4 5}; do
    if [ $3 = 128 ] && { [ $cfg = 0 ] || [ $cfg = 4 ]; }; then continue; fi
    timeout -k 10 120 python tools/conv_probe.py --halo 1 --rounds 2 --iters 10 --hw $1 --cin $2 --cout $3 --pro ${PRO:-1} --stats ${STATS:-1} --res ${RES:-0} --cfg $cfg --dtype ${DT:-bf16} | tail -1 || exit 1
  done
done
