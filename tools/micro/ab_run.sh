#!/bin/bash
# same-box A/B of two builds of the library: lib/abl_prev.so (the previous commit's) against lib/libnle_hip.so, alternating,
# cfg4: step time and the stage kernels' HIP-event times from bench.py's line
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in prev cur; do
    if [ $v = cur ]; then unset NLE_LIB_PATH; else export NLE_LIB_PATH=$GRAFT_REPO_ROOT/nonlocal-image-edit_amd/lib/abl_prev.so; fi
    python bench.py --config ${1:-cfg4} --no-cpu-baseline --no-pipelined --no-affinity --h2h-runs 0 --soak-seconds 0 --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; s=d['stage_ms_last_step']
print('$v','step %.3f ms'%d['ms_per_step'],'pass %.1f us'%(1e3*k['sinkhorn_pass']['avg_ms']),'sinkhorn stage %.3f'%s['sinkhorn'],'gram %.3f host %.3f'%(s['gram'],s['host']))
"
  done
done
