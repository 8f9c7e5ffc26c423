#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 --kernel-trace CSV: per kernel name calls / mean duration, and over the busy span
of the trace the share of wall time with 0, 1, 2, ... kernels in flight (do the graph replays on several streams overlap?)."""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = []
per = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    per.setdefault(name, []).append(e - s)
    ev.append((s, 1)); ev.append((e, -1))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5          # analyse the last `tail` fraction of the trace
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
cut = t1 - (t1 - t0) * tail
depth, last, hist = 0, None, {}
for t, d in ev:
    if last is not None and t > cut:
        hist[depth] = hist.get(depth, 0) + (t - max(last, cut))
    depth += d
    last = t
tot = sum(hist.values())
print("span analysed: %.2f ms (last %.0f %% of the trace)" % (tot / 1e6, tail * 100))
for k in sorted(hist):
    print("  %d kernels in flight: %5.1f %%" % (k, 100.0 * hist[k] / tot))
for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-60s %6d calls  mean %8.1f us  total %8.2f ms" % (name, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e6))
