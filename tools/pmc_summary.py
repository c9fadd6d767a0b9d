"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean value per dispatch)."""
import csv
import sys
from collections import defaultdict

out = defaultdict(lambda: [0, 0.0])
name = None
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
    out[(k, row["Counter_Name"])][0] += 1
    out[(k, row["Counter_Name"])][1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_value", "total_value"])
for (k, c), (n, s) in sorted(out.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, c, n, s / max(n, 1), s])
