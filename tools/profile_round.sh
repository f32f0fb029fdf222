#!/bin/bash
# Round profile: three rocprofv3 passes of the SAME bench command (kernel trace + stats, then the two PMC
# counters in passes of their own, as MI355X_MICROARCH.md prescribes), condensed by profiles/summarize.py.
#   usage (on the GPU box): bash tools/profile_round.sh r01      -> gpurun_out/prof_r01/{summary.txt,...}
set -e
TAG=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu --no-pmc --steps 3 --warmup 1 > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --no-cpu --no-pmc --steps 3 --warmup 1 > /dev/null 2> "$OUT/pmc_write.err"
python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt"
cp "$(ls "$OUT"/trace/*/*_kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
# keep what is merged back small: the raw traces stay on the box
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"
head -12 "$OUT/summary.txt"
