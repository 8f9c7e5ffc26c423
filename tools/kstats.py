#!/usr/bin/env python3
"""Print (kernel, calls, avg us, total ms) from a rocprofv3 *kernel_stats.csv (names contain commas)."""
import csv, sys, glob, os
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[-1]
pat = sys.argv[2:] 
for row in csv.DictReader(open(path)):
    name = row["Name"]
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    short = short.replace("(anonymous namespace)::", "")[:70]
    if pat and not any(p in name for p in pat):
        continue
    print("%-70s %6s calls  avg %10.1f us  total %9.2f ms" % (short, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
