"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: sum of each counter over dispatches."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:18.0f}   per-dispatch {v / n[(k, c)]:16.0f}  (n={n[(k, c)]})")
