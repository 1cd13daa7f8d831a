for m in 1 2 4; do NLE_SORTED_WGS_PER_CU=$m timeout -k 10 120 python bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 10 --warmup 3 > /tmp/w.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/w.json')); print('wgs/cu $m', round(d['ms_per_step'],3), {k: round(x['avg_ms']*1e3,1) for k,x in d['kernels'].items()})"; done
