#!/bin/bash
# per-kernel averages with the column factors by recurrence (default) and all from the LDS table (NLE_SORTED_TABLE=1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in cfg4 cfg3 cfg5 cfg2; do
  for t in rec table; do
    if [ $t = table ]; then export NLE_SORTED_TABLE=1; else unset NLE_SORTED_TABLE; fi
    timeout -k 10 200 python $ROOT/bench.py --config $cfg --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 5 --warmup 2 > /tmp/rc.json 2> /tmp/rc.err
    python - <<PY
import json
try:
    d=json.load(open("/tmp/rc.json")); print("$cfg $t", "ms/step %.3f" % d["ms_per_step"], {k: round(x["avg_ms"]*1e3,1) for k,x in d["kernels"].items()})
except Exception as e:
    print("$cfg $t failed", e); print(open("/tmp/rc.err").read()[-300:])
PY
  done
done
