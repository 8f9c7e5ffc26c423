#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel from the counter_collection CSV(s) under a directory.
usage: pmc_summary.py DIR [substring ...]   -> kernel, dispatches, counter = mean per dispatch"""
import csv, glob, os, sys, collections
d = sys.argv[1]
pats = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        if pats and not any(p in row["Kernel_Name"] for p in pats):
            continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[name].add(row["Dispatch_Id"])
for name in acc:
    n = max(len(cnt[name]), 1)
    print("%-60s %5d dispatches  " % (name, n) + "  ".join("%s=%.4g" % (k, v / n) for k, v in sorted(acc[name].items())))
