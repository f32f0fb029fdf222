#!/bin/bash
# Issued vs useful MFMA of the batched IVF tile scan (1M x 768, nlist 1024, nprobe 32, batch 1024): one rocprofv3 --pmc
# pass per counter (kernel trace only beside it), summarised by tools/pmc_summary.py.
#   usage (on the GPU box): bash tools/mfma_pmc.sh [tag]      -> gpurun_out/mfma_<tag>.txt
set -e
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/mfma_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT.txt"
for CTR in SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA; do
  rm -rf "$OUT/$CTR"
  if rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$OUT/$CTR" -- python3 tools/ivf_batch_time.py 1024 > "$OUT/$CTR.log" 2>&1; then
    echo "== $CTR" >> "$OUT.txt"
    python3 tools/pmc_summary.py "$OUT/$CTR" tile_scan_kernel >> "$OUT.txt" 2>&1 || echo "  (no rows)" >> "$OUT.txt"
  else
    echo "== $CTR: rocprofv3 failed" >> "$OUT.txt"; tail -3 "$OUT/$CTR.log" >> "$OUT.txt"
  fi
  rm -rf "$OUT/$CTR"
done
cat "$OUT.txt"
