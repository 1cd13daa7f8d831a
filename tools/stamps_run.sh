#!/bin/bash
# phase timestamps of k_sorted_pass for two co-resident workgroup pairs (tools/abl_build.sh STAMPS builds
# lib/abl_STAMPS.so, loaded through NLE_LIB_PATH: the product library is never touched)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/nonlocal-image-edit_amd/lib
NLE_LIB_PATH=$L/abl_STAMPS.so timeout -k 10 120 python $ROOT/bench.py --no-cpu-baseline --no-pipelined --h2h-runs 0 --steps 1 --warmup 1 2>&1 | grep STAMP | tail -64
