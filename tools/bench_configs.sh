#!/bin/bash
# usage: tools/bench_configs.sh [cfgs...]   (on the GPU box) -- ms per step of bench.py on each config, GPU leg only
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for c in "${@:-cfg3 cfg5}"; do
  for cc in $c; do
    python3 $ROOT/bench.py --config $cc --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined --h2h-runs 0 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cc', '${NLE_HOST_SOLVER:+host-solver}', round(d['ms_per_step'], 2), 'ms/step', d.get('stage_ms_last_step'))"
  done
done
