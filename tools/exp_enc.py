import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels
from tests import datagen
from bench import random_weights
from svision_amd.network.alexnet import AlexNet
dev = torch.device("cuda:0")
n = 4096
rec = torch.from_numpy(datagen.random_records(n, seed=3, hostile=False)).to(dev)
net = AlexNet(random_weights(0), device=dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timed(lambda: kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base))
print("SVX_ENC_DEBUG=%s: %.1f us per 64 images" % (os.environ.get("SVX_ENC_DEBUG", "0"), t / n * 64 * 1e3))
