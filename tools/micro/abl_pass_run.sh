#!/bin/bash
# ablations of k_sorted_pass (scratch builds under lib/abl_*.so): pass kernel time per variant and config
cd $GRAFT_REPO_ROOT
for c in cfg4 cfg5 cfg3; do
  for v in "" NOLOOP NOTREE NOSTORE NOTREENOSTORE NOLOOPNOTREENOSTORE; do
    if [ -z "$v" ]; then unset NLE_LIB_PATH; else export NLE_LIB_PATH=$GRAFT_REPO_ROOT/nonlocal-image-edit_amd/lib/abl_$v.so; fi
    python bench.py --config $c --no-cpu-baseline --no-pipelined --no-affinity --h2h-runs 0 --soak-seconds 0 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$c','${v:-FULL}','pass %.1f us'%(1e3*k['sinkhorn_pass']['avg_ms']),'apply_reduce %.1f'%(1e3*k['apply_reduce']['avg_ms']))
"
  done
done
