#!/bin/bash
# sweep the conv3x3 shapes of the 'standard' UNet at 512x512 over the halo tile configs (run on the GPU box)
for shape in "512 128 128" "512 256 256" "512 384 128" "256 256 256" "256 512 256" "256 128 256" "128 256 256" "128 512 512" "128 768 256" "64 512 512" "64 1024 512" "64 256 512"; do
  set -- $shape
  for cfg in 0 1 2; do
    timeout -k 10 120 python tools/conv_probe.py --halo 1 --rounds 2 --iters 10 --hw $1 --cin $2 --cout $3 --pro 1 --stats 1 --res ${RES:-0} --cfg $cfg | tail -1 || exit 1
  done
done
