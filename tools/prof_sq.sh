#!/bin/bash
# usage: tools/prof_sq.sh <tag>  -- SQ issue / wait counters of the default bench (where do the sorted kernels' wave cycles go?)
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 2 --warmup 1 > $OUT/pmc_sq2.log 2>&1 || { tail -20 $OUT/pmc_sq2.log; exit 1; }
find $OUT -name '*counter_collection.csv' | head
