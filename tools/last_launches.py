"""Developer tool: print the last N kernel launches of a rocprofv3 --kernel-trace CSV directory in time order
(start offset and duration in microseconds) -- one search's launch sequence with its gaps.
usage: python tools/last_launches.py <trace_dir> [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    grid = r.get("Grid_Size") or r.get("Grid_Size_X")
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X")
    print("%-64s grid %9s wg %5s start %8.1f dur %8.1f us" % (
        r["Kernel_Name"].replace("void ", "")[:64], grid, wg, (int(r["Start_Timestamp"]) - t0) / 1e3,
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
