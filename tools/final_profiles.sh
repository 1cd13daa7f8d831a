set -o pipefail
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err || { tail -5 gpurun_out/r3_final_bench.err; exit 1; }
echo "bench cfg4 done"
python3 bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r3_cfg3_bench.json 2>/dev/null && echo "bench cfg3 done"
python3 bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r3_cfg5_bench.json 2>/dev/null && echo "bench cfg5 done"
bash tools/prof.sh r3_final > gpurun_out/prof_r3_final.log 2>&1 && echo "prof final done"
bash tools/prof.sh r3_cfg3 --config cfg3 > gpurun_out/prof_r3_cfg3.log 2>&1 && echo "prof cfg3 done"
bash tools/prof.sh r3_cfg5 --config cfg5 --steps 5 --warmup 2 > gpurun_out/prof_r3_cfg5.log 2>&1 && echo "prof cfg5 done"
du -sh gpurun_out
