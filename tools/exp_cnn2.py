#!/usr/bin/env python3
"""GPU experiment: fused-epilogue AlexNet vs torch ops (ms per batch of 64, 1 and 3 streams via DeviceStage)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd import kernels
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage
from tests import datagen
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("TORCH_BENCH", "0") == "1"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
p = random_weights(0)
B = 64
rec = torch.from_numpy(datagen.random_records(B*16, seed=1, hostile=False)).to(dev)
img = kernels.rasterize(rec[:B], layout="NCHW")
for fused in (True,):
    net = AlexNet(p, device=dev, fused=fused)
    print(f"fused={fused} eager 1 stream: {timeit(lambda: net.predict(img)):.3f} ms/batch", flush=True)
    ref = AlexNet(p, device=dev, fused=False).predict(img)[2]
    print("  max |softmax diff| vs torch ops:", float((net.predict(img)[2]-ref).abs().max()))
    print(f"  eager predict_records: {timeit(lambda: net.predict_records(rec[:B])):.3f} ms/batch")
    print("  max |softmax diff| sparse vs dense:", float((net.predict_records(rec[:B])[2]-ref).abs().max()))
    for ns, sp in ((1, False), (1, True), (2, True), (3, True), (4, True)):
        st = DeviceStage(net, B, dev, n_streams=ns, sparse_first_layer=sp)
        out = torch.empty((B*16, 6), device=dev)
        ms = timeit(lambda: st.run(rec, out), n=10, warm=2) / 16
        print(f"  graph x{ns} streams sparse={sp}: {ms:.3f} ms/batch -> {1.4407*B/ms:.1f} TFLOP/s", flush=True)
