#!/usr/bin/env python3
"""Latency of a tiny kernel + read-back on its own stream while 4 streams are saturated with CNN graph replays."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import random_weights
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage
from tests import datagen
dev = torch.device("cuda:0")
net = AlexNet(random_weights(0), device=dev)
st = DeviceStage(net, 64, dev, n_streams=4)
rec = torch.from_numpy(datagen.random_records(64 * 40, seed=1, hostile=False)).to(dev)
out = torch.empty((64 * 40, 6), device=dev)
side = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
x = torch.zeros(1 << 16, device=dev)
pinned = torch.empty(1 << 16, pin_memory=True)


def probe(kind):
    t = time.perf_counter()
    if kind == "cpu":
        with torch.cuda.stream(side):
            y = x + 1
            y.cpu()
    elif kind == "cpu_hi":
        with torch.cuda.stream(hi):
            y = x + 1
            y.cpu()
    elif kind == "pinned":
        with torch.cuda.stream(side):
            y = x + 1
            pinned.copy_(y, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        ev.synchronize()
    elif kind == "pinned_hi":
        with torch.cuda.stream(hi):
            y = x + 1
            pinned.copy_(y, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        ev.synchronize()
    elif kind == "kernel_only":
        with torch.cuda.stream(side):
            y = x + 1
            ev = torch.cuda.Event(); ev.record()
        ev.synchronize()
    return (time.perf_counter() - t) * 1e3


for kind in ("cpu", "cpu_hi", "pinned", "pinned_hi", "kernel_only"):
    idle = [probe(kind) for _ in range(20)]
    torch.cuda.synchronize()
    lat = []
    for rep in range(6):
        done = st.run(rec, out, after=None) if False else None
        ev = torch.cuda.Event(); ev.record()
        st.run(rec, out, after=ev)                       # 40 batches queued on 4 streams (~20 ms of device work)
        time.sleep(0.002)
        for _ in range(4):
            lat.append(probe(kind))
        torch.cuda.synchronize()
    print("%-12s idle %.3f ms   loaded: median %.3f  max %.3f ms" % (kind, np.median(idle), np.median(lat), max(lat)), flush=True)
