"""rocprofv3 --pmc CSV directory -> per-kernel means of every counter (summed over dimensions per dispatch).  argv: dir [name filter]"""
import csv, glob, sys, collections, json
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k, json.dumps({c: round(sum(v.values()) / len(v), 1) for c, v in cs.items()}), "dispatches", len(next(iter(cs.values()))))
