b() { python bench.py --no-cpu-baseline --no-modes --steps 10 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['roofline']['avg_ms'])"; }
b --dtype mixed
b --dtype mixed --opt 15=300
b --dtype mixed --opt 15=600
b --dtype mixed --opt 15=1000
b --dtype mixed --opt 1=7
b --dtype mixed --opt 1=7,15=400
b --dtype mixed
b --dtype bf16
b --dtype bf16 --opt 14=300
b --dtype bf16 --opt 14=600
b --dtype bf16
