import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:50s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
