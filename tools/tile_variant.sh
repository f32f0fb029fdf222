#!/bin/bash
# Diagnostic builds of the library with compile-time variants of the tile kernel:
#   HG_TILE_ABLATE bit mask (parts of the K-step removed): 1 = no MFMA, 2 = no LDS reads, 4 = no LDS stores,
#                  8 = no global loads in the loop, 16 = no epilogue

# usage: tools/tile_variant.sh name=-DHG_TILE_ABLATE=14 ...  ->  build_dbg/libhnswgpu_<name>.so
#        then HNSWGPU_LIBRARY=build_dbg/libhnswgpu_<name>.so python tools/tile_ablate.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  ( /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -c hnsw-clj_amd/csrc/engine.hip -o build_dbg/engine_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_dbg/libhnswgpu_$name.so build_dbg/engine_$name.o \
      hnsw-clj_amd/csrc/ivf.o hnsw-clj_amd/csrc/hnsw.o hnsw-clj_amd/csrc/persist.o ) &
done
wait
