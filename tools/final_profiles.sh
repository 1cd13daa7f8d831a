#!/bin/bash
# usage (on the GPU box, one part per gpurun call: each takes 5-15 min):
#   bash tools/final_profiles.sh <round tag, e.g. r4> bench|prof4|prof35|extra
# bench : the default driver line (cfg4) and the cfg3 / cfg5 lines            -> gpurun_out/<tag>_{final,cfg3,cfg5}_bench.json
# prof4 : rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of cfg4 (incl. the stand-alone affinity legs) -> gpurun_out/prof_<tag>_final
# prof35: the same for cfg3 and cfg5                                           -> gpurun_out/prof_<tag>_cfg3, _cfg5
# extra : materialised-mode profiles, timelines, dense-solver timing, determinism, soak
set -o pipefail
TAG=${1:-r4}; PART=${2:-bench}
cd $GRAFT_REPO_ROOT
case $PART in
bench)
  python3 bench.py > gpurun_out/${TAG}_final_bench.json 2> gpurun_out/${TAG}_final_bench.err || { tail -5 gpurun_out/${TAG}_final_bench.err; exit 1; }
  echo "bench cfg4 done"
  python3 bench.py --config cfg3 --no-cpu-baseline > gpurun_out/${TAG}_cfg3_bench.json 2>/dev/null && echo "bench cfg3 done"
  python3 bench.py --config cfg5 --no-cpu-baseline > gpurun_out/${TAG}_cfg5_bench.json 2>/dev/null && echo "bench cfg5 done"
  python3 bench.py --config cfg2 --no-cpu-baseline > gpurun_out/${TAG}_cfg2_bench.json 2>/dev/null && echo "bench cfg2 done"
  ;;
prof4)
  bash tools/prof.sh ${TAG}_final > gpurun_out/prof_${TAG}_final.log 2>&1 && echo "prof final done"
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_final ${TAG}_final_tmp >/dev/null 2>&1
  ;;
prof35)
  bash tools/prof.sh ${TAG}_cfg3 --config cfg3 > gpurun_out/prof_${TAG}_cfg3.log 2>&1 && echo "prof cfg3 done"
  bash tools/prof.sh ${TAG}_cfg5 --config cfg5 --steps 5 --warmup 2 > gpurun_out/prof_${TAG}_cfg5.log 2>&1 && echo "prof cfg5 done"
  ;;
extra)
  bash tools/prof_materialised.sh ${TAG}_mat_f32 > gpurun_out/prof_${TAG}_mat_f32.log 2>&1 && echo "materialised f32 done"
  NLE_NYSTROM_BF16X3=1 bash tools/prof_materialised.sh ${TAG}_mat_bf16x3 > gpurun_out/prof_${TAG}_mat_bf16x3.log 2>&1 && echo "materialised bf16x3 done"
  for c in cfg4 cfg3 cfg5; do bash tools/timeline_run.sh $c 300 > gpurun_out/${TAG}_timeline_$c.txt 2>&1; done; echo "timelines done"
  SIZES=200:50,300:50,400:50,500:100,600:100,700:100,800:100,900:100,1152:100 python3 tools/dense_solver_timing.py > gpurun_out/${TAG}_dense_solver_timing.txt 2>&1; echo "dense timing done"
  for c in cfg4 cfg3 cfg5; do echo "== $c"; python3 tools/determinism_check.py $c; done > gpurun_out/${TAG}_determinism.txt 2>&1; echo "determinism done"
  python3 tools/soak.py > gpurun_out/${TAG}_soak.json 2>gpurun_out/${TAG}_soak.err; echo "soak done"
  ;;
esac
du -sh gpurun_out
