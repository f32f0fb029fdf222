#!/bin/bash
# Hardware counters of the IVF bounds pass at one batch size (one rocprofv3 --pmc pass per counter, kernel trace only beside it).
#   usage (GPU box): bash tools/bounds_pmc.sh <nq> [kernel name pattern]     -> gpurun_out/bounds_pmc_<nq>.txt
NQ=${1:-16384}
PAT=${2:-stream_bounds}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/bounds_pmc_$NQ.txt
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT"
for CTR in SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_DATA_FIFO_FULL TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_INSTS_VALU; do
  D=/tmp/bpmc_$CTR
  rm -rf "$D"
  if rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$D" -- python3 tools/ivf_one.py cosine $NQ 6 > /tmp/bpmc.log 2>&1; then
    echo "== $CTR" >> "$OUT"
    python3 tools/pmc_summary.py "$D" "$PAT" >> "$OUT" 2>&1 || echo "  (no rows)" >> "$OUT"
  else
    echo "== $CTR: rocprofv3 failed" >> "$OUT"; tail -2 /tmp/bpmc.log >> "$OUT"
  fi
  rm -rf "$D"
done
cat "$OUT"
