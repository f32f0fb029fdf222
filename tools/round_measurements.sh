#!/bin/bash
# The measurement set behind profiles/rNN_ivf_batch_times.txt, _survivors.txt, _stamps.txt, _latency.txt (GPU box; the
# stamps need tools/build_stamps.sh first).  usage: bash tools/round_measurements.sh   -> gpurun_out/r03_*.txt
cd "$(dirname "$0")/.."
{
echo "## tools/ivf_batch_time.py, cosine, default configuration (survivor stream at every batch size)"
timeout -k 10 300 python tools/ivf_batch_time.py 1 8 32 64 128 256 512 1024 2048 4096 8192 16384 2>&1 | grep "batch\|lists"
echo "## the same batches on the MFMA tile scan (HNSWGPU_TILE_PAIRS=12: the boundary of handles without int8 rows)"
HNSWGPU_TILE_PAIRS=12 timeout -k 10 300 python tools/ivf_batch_time.py 1024 2048 4096 8192 16384 2>&1 | grep "batch"
echo "## Euclidean"
METRIC=l2 timeout -k 10 300 python tools/ivf_batch_time.py 1 32 256 1024 4096 2>&1 | grep "batch\|lists"
echo "## without the half-precision pass (HNSWGPU_TUNE=STREAM_MID=0)"
HNSWGPU_TUNE=STREAM_MID=0 timeout -k 10 300 python tools/ivf_batch_time.py 64 256 1024 2>&1 | grep "batch"
echo "## (the ablations of the wide epilogue -- results wrong on purpose -- need the -DHG_DIAG build: tools/build_stamps.sh + hnswgpu_debug_set_ablation)"
} > gpurun_out/r03_ivf_batch_times.txt 2>&1
{
timeout -k 10 200 python tools/ivf_survivors.py cosine 32 256 1024 4096 2>&1 | grep batch
timeout -k 10 200 python tools/ivf_survivors.py l2 32 256 1024 4096 2>&1 | grep batch
echo "## HNSWGPU_TUNE=STREAM_MID=0"
HNSWGPU_TUNE=STREAM_MID=0 timeout -k 10 200 python tools/ivf_survivors.py cosine 256 1024 2>&1 | grep batch
} > gpurun_out/r03_ivf_survivors.txt 2>&1
HNSWGPU_LIBRARY=build_dbg/libhnswgpu_stamps.so timeout -k 10 300 python tools/ivf_phase_stamps.py 1 8 > gpurun_out/r03_ivf_stamps.txt 2>&1
timeout -k 10 300 python tools/ivf_latency.py > gpurun_out/r03_ivf_latency.txt 2>&1
tail -5 gpurun_out/r03_ivf_latency.txt
