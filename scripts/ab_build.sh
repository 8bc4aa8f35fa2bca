#!/bin/bash
# Developer tool: build a variant of the library with extra -D switches into pc-gnn_amd/lib/ab/<name>.so
# (picked up with PCG_LIB=<path>).   usage: scripts/ab_build.sh <name> [-DPCG_X=1 ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
D=$R/pc-gnn_amd/lib/ab/$name; mkdir -p $D
objs=""
for s in score mark sort segmean_pick choose select gather dense halo; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-function "$@" -c $R/pc-gnn_amd/csrc/$s.hip -o $D/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/pc-gnn_amd/lib/ab/$name.so $D/*.o
rm -rf $D
echo $R/pc-gnn_amd/lib/ab/$name.so
