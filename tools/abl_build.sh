#!/bin/bash
# usage: tools/abl_build.sh V1 V2 ...   (V in NOLOOP NOTREE NOPRIO STAMPS; run where hipcc is, e.g. the build container)
# Builds measurement variants of the library as lib/abl_<V>.so: a COPY of csrc/ with tools/micro/sorted_pass_ablation.patch
# applied (the hooks of profiles/r2_pass_ablation.txt) compiled with -DNLE_ABL_<V>.  The product sources and
# lib/libnle_hip.so are never touched; tools/abl_run.sh / tools/stamps_run.sh load the variants through NLE_LIB_PATH.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PKG=$ROOT/nonlocal-image-edit_amd
TMP=$(mktemp -d /tmp/nle_abl.XXXXXX)
cp -r $PKG/csrc $TMP/csrc
patch -s -p1 -d $TMP/csrc < $ROOT/tools/micro/sorted_pass_ablation.patch
for v in "$@"; do
  objs=""
  for f in kernels tsgemm_bf16x3 fused sorted generic64 tridiag dense64 colour pipeline ortho abi_ctx devsolve; do
    /opt/rocm/bin/hipcc -x hip -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DNLE_ABL_$v -I $ROOT/include -c $TMP/csrc/$f.hip -o $TMP/$f.$v.o &
    objs="$objs $TMP/$f.$v.o"
  done
  g++ -O3 -std=c++17 -fPIC -I $ROOT/include -c $TMP/csrc/eigen_sym.cpp -o $TMP/eigen_sym.$v.o &
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/lib/abl_$v.so $objs $TMP/eigen_sym.$v.o -lpthread -ldl
  echo "built $PKG/lib/abl_$v.so"
done
rm -rf $TMP
