#!/bin/bash
# Diagnostic build: the product objects + solo.hip with cycle stamps in the sequencer (-DHG_SOLO_STAMPS; tools/sq_probe.py prints them)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
make -C hnsw-clj_amd/csrc -j6 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DHG_SOLO_STAMPS \
  -c hnsw-clj_amd/csrc/solo.hip -o build_dbg/solo_stamps.o
cd hnsw-clj_amd/csrc
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_dbg/libhnswgpu_solostamps.so engine.o ivf.o hnsw.o wave.o persist.o group.o ../../build_dbg/solo_stamps.o
