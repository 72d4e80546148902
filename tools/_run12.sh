mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
o=gpurun_out/r3c
python bench.py --config c1 --graph --steps 50 --warmup 5 > $o/bench_c1.json 2> $o/c1.err; tail -c 300 $o/bench_c1.json; echo
python bench.py --config c2 --graph --steps 30 --warmup 5 > $o/bench_c2.json 2> $o/c2.err; tail -c 300 $o/bench_c2.json; echo
python bench.py --config c3 > $o/bench_c3.json 2> $o/c3.err; tail -c 300 $o/bench_c3.json; echo
python bench.py --config c4 > $o/bench_c4.json 2> $o/c4.err; tail -c 300 $o/bench_c4.json; echo
for c in c1 c2 c3 c4; do
  extra=""; [ $c = c1 ] && extra="--steps 30 --warmup 5"; [ $c = c2 ] && extra="--steps 20 --warmup 3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_$c -- python3 bench.py --config $c --no-cpu-baseline --no-modes --no-kernel-events $extra > $o/bench_${c}_under_rocprof.json 2> $o/stats_$c.err
  s=$(find $o/stats_$c -name "*kernel_stats.csv" | head -1); cp $s $o/bench_${c}_kernel_stats.csv; rm -rf $o/stats_$c
  echo "stats $c done"
done
python bench.py --config c3 --backward --no-cpu-baseline --steps 10 > $o/bench_c3_backward.json 2> $o/c3b.err; tail -c 400 $o/bench_c3_backward.json; echo
python bench.py --config c5-noclip --backward --no-cpu-baseline --steps 10 > $o/bench_c5_unet_backward.json 2> $o/c5b.err; tail -c 400 $o/bench_c5_unet_backward.json; echo
python bench.py --config c2 --backward --no-cpu-baseline --steps 10 > $o/bench_c2_backward.json 2> $o/c2b.err; tail -c 300 $o/bench_c2_backward.json; echo
tail -3 $o/c5b.err
