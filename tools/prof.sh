#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -- rocprofv3 kernel-trace stats + HBM PMC passes of bench.py
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --soak-seconds 0 "$@" > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --soak-seconds 0 "$@" > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --soak-seconds 0 "$@" > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
find $OUT -name '*.csv' | head -30
