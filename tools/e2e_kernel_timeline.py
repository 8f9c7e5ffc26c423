#!/usr/bin/env python3
"""The file-inclusive leg of `bench.py --no-other-engine --no-cpu-baseline --no-calibration` in a rocprofv3 --kernel-trace CSV:
which kernels the device ran, class by class (union of their intervals = time with at least one kernel of the class in flight;
sum = kernel time), when nothing ran, and how long the ingest kernels ran next to the CNN.  The leg is found by its tokens
launches: the last cluster of bgzf_tokens_kernel launches of the trace (the warm-up pass from the file comes before it), or the
last argv[2] of them."""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
tok = [r for r in rows if "bgzf_tokens_kernel" in r[2]]
if len(sys.argv) > 2:                                           # argv[2]: the leg's number of inflate launches (7 for the 20-window job)
    cluster = tok[-int(sys.argv[2]):]
else:
    cluster = [tok[-1]]
    for r in reversed(tok[:-1]):
        if cluster[0][0] - r[1] > 100e6:
            break
        cluster.insert(0, r)
lz = [r for r in rows if "bgzf_lz_kernel" in r[2] and r[0] >= cluster[0][0]]
t_start = cluster[0][0] - 20e6                               # read + index of the first group in front of its launch
cnn = ("conv_wave", "fc_splitk", "fc_reduce", "encode_conv1", "bias_relu_pool", "active_", "fc8_softmax")
after = [r for r in rows if r[0] >= t_start]
# the leg ends where the CNN pauses for the resident leg's set-up: the first gap of > 25 ms behind the last LZ launch
t_end, last = None, lz[-1][1]
for s, e, n in after:
    if s < lz[-1][1]:
        last = max(last, e)
        continue
    if s - last > 25e6:
        t_end = last
        break
    last = max(last, e)
t_end = t_end or last
leg = [(max(s, t_start), min(e, t_end), n) for s, e, n in after if s < t_end]
def klass(n):
    if "bgzf_tokens" in n: return "tokens"
    if "bgzf_lz" in n: return "lz"
    if "bgzf_crc" in n or "bam_walk" in n or "bgzf_inflate_wave" in n: return "crc + walk"
    if any(k in n for k in ("count_kernel", "offsets_kernel", "emit_kernel")): return "cigar scan"
    if any(k in n for k in cnn): return "cnn"
    return "other (copies, fills, torch)"
def union(iv):
    iv = sorted(iv); tot, lo, hi = 0, None, None
    for s, e in iv:
        if hi is None or s > hi:
            if hi is not None: tot += hi - lo
            lo, hi = s, e
        else:
            hi = max(hi, e)
    return tot + (hi - lo if hi is not None else 0)
by = {}
for s, e, n in leg:
    by.setdefault(klass(n), []).append((s, e))
span = t_end - t_start
print("leg: %.1f ms from 20 ms in front of the first tokens launch to the last kernel (%d tokens launches)" % (span / 1e6, len(cluster)))
for k in ("cnn", "tokens", "lz", "crc + walk", "cigar scan", "other (copies, fills, torch)"):
    iv = by.get(k, [])
    print("  %-30s %4d launches  in flight %6.1f ms  kernel time %7.1f ms" % (k, len(iv), union(iv) / 1e6, sum(e - s for s, e in iv) / 1e6))
allu = union([(s, e) for s, e, _n in leg])
print("  any kernel in flight %.1f ms, nothing in flight %.1f ms" % (allu / 1e6, (span - allu) / 1e6))
def overlap(a, b):
    ev = [(s, 1, 0) for s, e in a] + [(e, -1, 0) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, -1, 1) for s, e in b]
    ev.sort()
    d = [0, 0]; last = None; tot = 0
    for t, x, w in ev:
        if last is not None and d[0] > 0 and d[1] > 0: tot += t - last
        d[w] += x; last = t
    return tot
print("  cnn next to tokens %.1f ms, cnn next to lz %.1f ms, cnn alone %.1f ms" % (
    overlap(by.get("cnn", []), by.get("tokens", [])) / 1e6, overlap(by.get("cnn", []), by.get("lz", [])) / 1e6,
    (union(by.get("cnn", [])) - overlap(by.get("cnn", []), by.get("tokens", []) + by.get("lz", []))) / 1e6))
for i, (s, e, _n) in enumerate(cluster):
    l = [r for r in lz if r[0] >= s][:1]
    print("  group %d: tokens %.1f -> %.1f ms (%.1f), lz -> %.1f ms (%.1f)" % (i + 1, (s - t_start) / 1e6, (e - t_start) / 1e6, (e - s) / 1e6,
          (l[0][1] - t_start) / 1e6 if l else -1, (l[0][1] - l[0][0]) / 1e6 if l else -1))
