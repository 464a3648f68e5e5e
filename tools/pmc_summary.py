"""Per-kernel average of one rocprofv3 PMC counter (counter_collection.csv -> text table)."""
import collections
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        a = acc[row["Kernel_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
print(f"# {counter}: kernel, dispatches, sum, mean per dispatch (raw counter units as rocprofv3 reports them)")
for name, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{name[:150]}\t{n}\t{s:.6g}\t{s / n:.6g}")
