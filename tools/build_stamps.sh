#!/bin/bash
# Diagnostic build of the library with per-phase time stamps in the HNSW kernel and along an IVF search, and with the
# ablation switches of the bounds pass / tile scan (-DHG_DIAG: hnswgpu_debug_set_ablation; results are wrong on purpose) -- never the product build.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
FLAGS="-O3 -DHG_SOLO_STAMPS -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DHG_HNSW_STAMPS -DHG_IVF_STAMPS -DHG_DIAG"
for f in engine ivf hnsw solo wave persist group; do
  /opt/rocm/bin/hipcc $FLAGS -c hnsw-clj_amd/csrc/$f.hip -o build_dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_dbg/libhnswgpu_stamps.so build_dbg/{engine,ivf,hnsw,solo,wave,persist,group}.o
