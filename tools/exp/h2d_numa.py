"""Does the H2D rate of a pinned buffer depend on the NUMA node its pages sit on?  The buffer is allocated (and touched) by a
thread bound to one node's CPUs; then 64 MB copies are timed.  Also: where is the GPU?"""
import glob, os, sys, threading, time
import numpy as np, torch
torch.cuda.init()
dev = torch.device("cuda:0")
for p in glob.glob("/sys/class/drm/card*/device/numa_node") + glob.glob("/sys/bus/pci/devices/*/numa_node")[:0]:
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node*")):
    try:
        cpus = open(os.path.join(d, "cpulist")).read().strip()
    except OSError:
        continue
    ids = []
    for part in cpus.split(","):
        if "-" in part:
            a, b = part.split("-"); ids += list(range(int(a), int(b) + 1))
        elif part:
            ids.append(int(part))
    nodes[os.path.basename(d)] = ids
    print(os.path.basename(d), cpus)
allowed = os.sched_getaffinity(0)
dst = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def rate(pin, reps=8):
    dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    return reps * pin.numel() / (time.perf_counter() - t) / 1e9


for name, cpus in nodes.items():
    use = sorted(set(cpus) & allowed)
    if not use:
        continue
    out = {}

    def work():
        os.sched_setaffinity(0, use)
        pin = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
        pin.numpy()[:] = 1
        out["r"] = rate(pin)
    th = threading.Thread(target=work); th.start(); th.join()
    print("pinned on %s (%d cpus): H2D %.1f GB/s" % (name, len(use), out["r"]), flush=True)
