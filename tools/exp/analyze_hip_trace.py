import csv, glob, collections
api = glob.glob("/tmp/rp_hip/**/*hip_api_trace.csv", recursive=True)[0]
ker = glob.glob("/tmp/rp_hip/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(api)))
print(rows[0].keys())
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
longs = [r for r in rows if r["Function"] in ("hipMemcpyWithStream", "hipEventSynchronize", "hipLaunchKernel", "hipMalloc", "hipHostMalloc", "hipMemcpyAsync") and E(r) - S(r) > 40e6]
longs.sort(key=S)
t_end = max(E(r) for r in rows)
by_tid = collections.defaultdict(list)
for r in rows: by_tid[r["Thread_Id"]].append(r)
for r in longs:
    tid = r["Thread_Id"]
    lst = by_tid[tid]
    i = lst.index(r)
    prev = [x["Function"] for x in lst[max(0, i - 6):i]]
    print("%-22s dur %.1f ms  start %.3f s before end  tid %s  prev: %s" % (r["Function"], (E(r) - S(r)) / 1e6, (t_end - S(r)) / 1e9, tid, prev))
krows = list(csv.DictReader(open(ker)))
print(krows[0].keys())
# GPU activity during each long call: kernels overlapping it
for r in longs:
    if r["Function"] != "hipMemcpyWithStream": continue
    s, e = S(r), E(r)
    ov = [(k["Kernel_Name"][:50], (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e6) for k in krows if int(k["Start_Timestamp"]) < e and int(k["End_Timestamp"]) > s]
    tot = sum(d for _n, d in ov)
    c = collections.Counter()
    for n, d in ov: c[n] += d
    print("during the %.0f ms memcpy at -%.3f s: %d kernels, %.0f ms of kernel time:" % ((e - s) / 1e6, (t_end - s) / 1e9, len(ov), tot), c.most_common(5))
# GPU idle gaps and long kernels in the last 2.5 s of the run (the two file-driven legs)
ks = sorted(((int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"][:48]) for k in krows), key=lambda x: x[0])
t_last = max(e for _s, e, _n in ks)
ks = [k for k in ks if k[0] > t_last - 2.6e9]
busy_until = ks[0][1]
for s, e, n in ks:
    if s - busy_until > 25e6:
        print("GPU idle %.0f ms until -%.3f s (next kernel: %s)" % ((s - busy_until) / 1e6, (t_last - s) / 1e9, n))
    busy_until = max(busy_until, e)
for s, e, n in ks:
    if e - s > 40e6:
        print("kernel %.0f ms at -%.3f s: %s" % ((e - s) / 1e6, (t_last - s) / 1e9, n))
