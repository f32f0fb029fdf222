"""Condense a rocprofv3 run (kernel trace CSV + optional PMC CSVs) into the per-kernel summary that
is committed under profiles/.  Dispatches are grouped by (kernel, grid size) because the scan kernel
serves several callers (k-means++ rounds, k-means assignment, centroid routing, list scan).

usage: python profiles/summarize.py <prof_dir> > profiles/rNN_summary.txt
PMC units: FETCH_SIZE / WRITE_SIZE are KB.  On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> the 'HBM read' column is 2 x FETCH_SIZE.
"""
import collections
import csv
import glob
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:56]


def main(d):
    groups = collections.OrderedDict()
    tr = glob.glob(d + "/trace/*/*_kernel_trace.csv")
    for f in tr:
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r.get("Grid_Size") or r["Grid_Size_X"]))
            g = groups.setdefault(key, {"n": 0, "ns": 0.0, "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size")})
            g["n"] += 1
            g["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(d + "/" + tag + "/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                pmc[(short(r["Kernel_Name"]), int(r.get("Grid_Size") or r["Grid_Size_X"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%-58s %10s %6s %12s %6s %7s %14s %14s" % ("kernel", "grid", "calls", "avg_us", "vgpr", "lds", "HBM read MB", "HBM write MB"))
    for (k, grid), g in sorted(groups.items(), key=lambda kv: -kv[1]["ns"]):
        if not k.startswith("hg::"):
            continue
        p = pmc.get((k, grid), {})
        fe = p.get("FETCH_SIZE")
        wr = p.get("WRITE_SIZE")
        rd = "%.2f" % (2 * sum(fe) / len(fe) * 1024 / 1e6) if fe else "-"
        ww = "%.2f" % (sum(wr) / len(wr) * 1024 / 1e6) if wr else "-"
        print("%-58s %10d %6d %12.1f %6s %7s %14s %14s" % (k, grid, g["n"], g["ns"] / g["n"] / 1e3, g["vgpr"], g["lds"], rd, ww))


if __name__ == "__main__":
    main(sys.argv[1])
