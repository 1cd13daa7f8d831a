#!/bin/bash
# usage: tools/prof_lds.sh <tag>  -- LDS activity counters of the default bench (the table kernels are LDS bound)
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $ROOT/bench.py --no-cpu-baseline --h2h-runs 0 --steps 2 --warmup 1 > $OUT/pmc_lds.log 2>&1 || { tail -20 $OUT/pmc_lds.log; exit 1; }
find $OUT -name '*.csv' | head
