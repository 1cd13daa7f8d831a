#!/bin/bash
# usage: tools/prof_materialised.sh <tag>  -- materialised formulation (NLE_MODE_MATERIALISED): kernel trace stats and an
# MFMA-activity PMC pass of bench.py --mode 1 (the affinity-fused Nystrom extension GEMM is k_tsgemm<7, true>)
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --mode 1 --steps 2 --warmup 1 > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/bench.py --no-cpu-baseline --mode 1 --steps 2 --warmup 1 > $OUT/pmc_mfma.log 2>&1 || { tail -20 $OUT/pmc_mfma.log; exit 1; }
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench.json   # the bench line, not rocprofv3's last log line
find $OUT -name '*.csv' | head
