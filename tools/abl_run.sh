#!/bin/bash
# usage: tools/abl_run.sh V1 V2 ...  -- per-kernel averages of the default bench with lib/abl_<V>.so in place of the library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/nonlocal-image-edit_amd/lib
cp $L/libnle_hip.so /tmp/libnle_hip.keep
for v in "$@"; do
  cp $L/abl_$v.so $L/libnle_hip.so
  timeout -k 10 120 python $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 10 --warmup 3 > /tmp/abl_$v.json 2> /tmp/abl_$v.err
  python - <<PY
import json
try:
    d=json.load(open("/tmp/abl_$v.json")); print("$v", "ms/step %.3f" % d["ms_per_step"], {k: round(x["avg_ms"]*1e3,1) for k,x in d["kernels"].items()})
except Exception as e:
    print("$v failed", e); print(open("/tmp/abl_$v.err").read()[-300:])
PY
done
cp /tmp/libnle_hip.keep $L/libnle_hip.so
