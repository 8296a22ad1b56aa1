"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of every counter."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for fn in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "assemble" not in k and "_kernel" not in k:
                continue
            acc[(k[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{k:72s} {c:24s} n={len(v)} mean={sum(v)/len(v):.6g}")
