"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel family.
usage: python tools/pmc_summary.py gpurun_out/pmc_x/<host>/<pid>_counter_collection.csv"""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    key = name.split("(")[0][:60]
    agg[(key, r["Counter_Name"])][0] += 1
    agg[(key, r["Counter_Name"])][1] += float(r["Counter_Value"])
print(f"{'kernel':60s} {'counter':12s} {'launches':>8s} {'sum':>16s} {'per launch':>14s}")
for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:60s} {c:12s} {n:8d} {v:16.1f} {v / n:14.1f}")
