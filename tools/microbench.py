#!/usr/bin/env python3
"""Streaming-rate microbenchmarks of the hand-written kernels at sizes that leave the caches
(run under rocprofv3 --kernel-trace --stats, and again with --pmc FETCH_SIZE / WRITE_SIZE)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels
from tests import datagen
from bench import random_weights
from svision_amd.network.alexnet import AlexNet
dev = torch.device("cuda:0")
reps = int(os.environ.get("REPS", "10"))
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
out = {}
# rasteriser: 4096 images = 2.53 GB written
n = 4096
rec = torch.from_numpy(datagen.random_records(n, seed=3, hostile=False)).to(dev)
for lay in ("NCHW", "NHWC"):
    buf = torch.empty(n * 3 * 227 * 227, dtype=torch.float32, device=dev)
    t = timed(lambda: kernels.rasterize(rec, layout=lay, out=buf))
    out["raster_" + lay] = {"images": n, "s": t, "GBps": n * 618348 / t / 1e9}
# CIGAR scan: 2M alignments x ~150 ops
na = int(os.environ.get("N_ALN", "2000000"))
cigar, off, ref_start = datagen.random_cigars(na, seed=5, mean_ops=150, long_gap_rate=0.0005)
d_c = torch.from_numpy(cigar.view(np.int32)).to(dev); d_o = torch.from_numpy(off.astype(np.int64)).to(dev); d_r = torch.from_numpy(ref_start).to(dev)
cap = 1 << 22
t = timed(lambda: kernels.cigar_scan(d_c, d_o, d_r, 50, gaps_cap=cap))
res = kernels.cigar_scan(d_c, d_o, d_r, 50, gaps_cap=cap)
out["cigar_scan"] = {"alignments": na, "ops": int(cigar.size), "gaps": res.total(), "s": t,
                     "GBps_algorithmic": (4 * cigar.size + 32 * na + 24 * res.total()) / t / 1e9}
# sparse first layer, 4096 images
net = AlexNet(random_weights(0), device=dev)
t = timed(lambda: kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base))
out["encode_conv1"] = {"images": n, "s": t, "us_per_64": t / n * 64 * 1e6}
x = kernels.to_c8(torch.randn(256, 96, 55, 55, device=dev)); b = torch.randn(96, device=dev)
t = timed(lambda: kernels.bias_relu_pool_lrn(x, b))
out["pool_lrn_1"] = {"s": t, "GBps": (x.numel() * 4 + 256 * 96 * 27 * 27 * 4) / t / 1e9}
print(json.dumps(out))
