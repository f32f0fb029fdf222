#!/bin/bash
# Per-kernel times of one IVF search per configuration (rocprofv3 kernel trace).
#   usage (GPU box): bash tools/ivf_trace.sh "cosine 32" "l2 1024" ...   -> gpurun_out/ivf_trace.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ivf_trace.txt
: > "$OUT"
for cfg in "$@"; do
    set -- $cfg
    D=/tmp/ivftrace_$1_$2
    rm -rf "$D"
    echo "== $cfg" >> "$OUT"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/ivf_one.py $cfg 2> /tmp/ivf_trace.err | tail -1 >> "$OUT"
    python3 - "$D" "${3:-50}" >> "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
reps = int(sys.argv[2]) + 5
rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= reps - 6 and int(r["Calls"]) <= 2 * reps + 12]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["AverageNs"])):
    per = float(r["TotalDurationNs"]) / reps / 1e3
    tot += per
    print("  %-70s calls %4s  avg %8.1f us   per search %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, per))
print("  sum per search %.1f us" % tot)
PY
done
cat "$OUT"
