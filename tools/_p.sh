set -e
python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'], d['roofline']['achieved'])"
python tools/sd_trace.py 2>/dev/null | grep "N=320 K=320\|N=640 K=5760\|N=640 K=11520\|N=640 K=17280\|N=640 K=8640\|N=320 K=1280\|N=640 K=640\|total"
