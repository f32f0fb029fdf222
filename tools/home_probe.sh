#!/bin/bash
# tools/home_probe.py under rocprofv3 for a list of tuning settings: average duration of ivf_home_kernel per setting.
#   usage (GPU box): bash tools/home_probe.sh "" "HOME_CHUNK=512" "HOME_DEPTH=4" ...  -> gpurun_out/home_probe.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/home_probe.txt
: > "$OUT"
for cfg in "$@"; do
    D=/tmp/homeprobe_$RANDOM
    rm -rf "$D"
    export TUNE="$cfg"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/home_probe.py ${NQ:-4} 10 > /tmp/home_probe.out 2> /tmp/home_probe.err
    python3 - "$D" "$cfg" >> "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "ivf_home_kernel" in r["Name"]:
        print("%-40s %s calls %s avg %.1f us min %.1f" % (sys.argv[2] or "(default)", r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
cat "$OUT"
