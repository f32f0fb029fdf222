#!/bin/bash
# Diagnostic builds of the library with parts of the tile kernel's K-step removed (HG_TILE_ABLATE bit mask:
# 1 = no MFMA, 2 = no LDS reads, 4 = no LDS stores, 8 = no global loads in the loop, 16 = no epilogue).
# usage: tools/tile_ablate.sh 1 2 4 ...   ->  build_dbg/libhnswgpu_abl<mask>.so (objects other than engine.o reused)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DHG_TILE_ABLATE=$m -c hnsw-clj_amd/csrc/engine.hip -o build_dbg/engine_abl$m.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_dbg/libhnswgpu_abl$m.so build_dbg/engine_abl$m.o \
      hnsw-clj_amd/csrc/ivf.o hnsw-clj_amd/csrc/hnsw.o hnsw-clj_amd/csrc/persist.o ) &
done
wait
