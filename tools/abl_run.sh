#!/bin/bash
# usage: tools/abl_run.sh V1 V2 ...  -- per-kernel averages of the default bench with the measurement build lib/abl_<V>.so (made by tools/abl_build.sh)
# loaded through NLE_LIB_PATH (the product library lib/libnle_hip.so is never touched)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/nonlocal-image-edit_amd/lib
for v in "$@"; do
  NLE_LIB_PATH=$L/abl_$v.so timeout -k 10 120 python $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 10 --warmup 3 > /tmp/abl_$v.json 2> /tmp/abl_$v.err
  python - <<PY
import json
try:
    d=json.load(open("/tmp/abl_$v.json")); print("$v", "ms/step %.3f" % d["ms_per_step"], {k: round(x["avg_ms"]*1e3,1) for k,x in d["kernels"].items()})
except Exception as e:
    print("$v failed", e); print(open("/tmp/abl_$v.err").read()[-300:])
PY
done
