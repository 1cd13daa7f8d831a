#!/bin/bash
# usage: tools/timeline_run.sh <cfg> [gap_us]   (on the GPU box) -- kernel timeline of one traced step of a config
set -o pipefail
CFG=${1:-cfg4}; GAP=${2:-50}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl_$CFG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/trace_run.py $CFG > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
python3 $ROOT/tools/timeline.py $OUT/trace $GAP > $OUT/timeline.txt
grep 'nle trace\|nle eig' $OUT/trace.log | tail -30 >> $OUT/timeline.txt
find $OUT/trace -name '*.csv' -delete
tail -50 $OUT/timeline.txt
