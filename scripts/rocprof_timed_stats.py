#!/usr/bin/env python3
"""Per-kernel duration statistics over the TIMED launches of a bench.py run profiled with `rocprofv3 --kernel-trace`:
the first WARMUP launches of every assemble_* kernel are dropped (cold launches: round 2's averages included them and came
out 3.7 % above the bench line's own event timing).
usage: rocprof_timed_stats.py DIR WARMUP > stats.csv     (DIR holds .../*kernel_trace.csv)"""
import collections
import csv
import glob
import sys

d, warm = sys.argv[1], int(sys.argv[2])
rows = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "WarmupDropped", "AverageNs", "MinNs", "MaxNs"])
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if "assemble_" not in name and "first_comp" not in name:
        continue
    t = v[warm:] if len(v) > warm else v
    w.writerow([name, len(t), len(v) - len(t), "%.1f" % (sum(t) / len(t)), min(t), max(t)])
