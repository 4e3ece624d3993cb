#!/usr/bin/env bash
# Diagnostic builds of a K3 kernel with one operand stream removed (results are garbage; timing only):
#   now = no weight loads, nob = no LDS activation loads (direct kernel only), nowb = neither; or any +-joined macro list
#   (split kernel: AZ_K3S_DBG_NOW, AZ_K3S_DBG_NOB, AZ_K3S_DBG_NOEPI).
# Output: dbgbuild/libazamd_<v>.so (git-ignored, travels with gpurun).
# Usage (build here, run on the GPU box):  AZAMD_LIB=dbgbuild/libazamd_now.so python tools/k3_sweep.py 1280 6400
set -euo pipefail
cd "$(dirname "$0")/../alphazero_amd/csrc"
OUT=../../dbgbuild
mkdir -p $OUT
for v in ${VARIANTS:-now}; do
  case $v in now) D="-DAZ_K3_DBG_NOW";; nob) D="-DAZ_K3_DBG_NOB";; nowb) D="-DAZ_K3_DBG_NOW -DAZ_K3_DBG_NOB";; *) D=$(echo $v | sed 's/+/ -D/g; s/^/-D/');; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I../../include $D -c az_cnn.hip -o $OUT/az_cnn_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -I../../include $D -c az_cnn_wino.hip -o $OUT/az_cnn_wino_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I../../include $D -c az_cnn_split.hip -o $OUT/az_cnn_split_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libazamd_$v.so build/az_tree.o build/az_play.o build/az_env.o build/az_traj.o build/az_engine.o $OUT/az_cnn_split_$v.o $OUT/az_cnn_$v.o $OUT/az_cnn_wino_$v.o
done
ls -la $OUT/*.so
