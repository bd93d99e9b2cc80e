"""mean per kernel of every counter in the given rocprofv3 counter_collection.csv files -> one CSV table"""
import csv, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in vals for c in vals[k]})
print(",".join(["kernel", "launches"] + counters))
for k in sorted(vals, key=lambda k: -len(next(iter(vals[k].values())))):
    n = max(len(v) for v in vals[k].values())
    print(",".join([k, str(n)] + [("%.4g" % (sum(vals[k][c]) / len(vals[k][c])) if c in vals[k] else "") for c in counters]))
