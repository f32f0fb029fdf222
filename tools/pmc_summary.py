import csv, glob, collections, sys
d=sys.argv[1]; pat=sys.argv[2]
f=glob.glob(d+"/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Grid_Size"]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for g,c in agg.items():
    n=len(next(iter(c.values())))
    print("grid",g,"calls",n,"avg dur us %.1f"%(sum(dur[g])/len(dur[g])/1e3), " ".join("%s=%.4g"%(k,sum(v)/len(v)) for k,v in sorted(c.items())))
