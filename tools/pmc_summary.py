"""Summarise a rocprofv3 counter_collection.csv: mean per dispatch of each counter per kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", r.get("Kernel Name", "?"))[:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
