#!/usr/bin/env python3
"""GPU experiment: own MFMA implicit-GEMM conv vs MIOpen, per layer and in-net."""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd import kernels
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage
from tests import datagen
dev = torch.device("cuda:0")
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
B=64
p = random_weights(0)
net = AlexNet(p, device=dev)
rec = torch.from_numpy(datagen.random_records(B*16, seed=1, hostile=False)).to(dev)
x1 = kernels.encode_conv1(rec[:B], net.conv1_hwio, net.conv1_base)           # realistic activations
acts = {"conv2": x1}
x = kernels.bias_relu_pool_lrn(F.conv2d(x1, net.conv2_w, None, 1, 2, 1, 2), net.conv2_b); acts["conv3"] = x
x = F.relu(F.conv2d(x, net.conv3_w, net.conv3_b, 1, 1)); acts["conv4"] = x
x = F.relu(F.conv2d(x, net.conv4_w, net.conv4_b, 1, 1, 1, 2)); acts["conv5"] = x
for name, g, k in (("conv2",2,5),("conv3",1,3),("conv4",2,3),("conv5",2,3)):
    xin = acts[name]; w_oihw = getattr(net, name+"_w"); w_hwio = getattr(net, name+"_hwio")
    gf = 2*B*xin.shape[2]*xin.shape[3]*w_hwio.shape[3]*w_hwio.shape[2]*k*k/1e9
    t_mi = timeit(lambda: F.conv2d(xin, w_oihw, None, 1, k//2, 1, g))
    t_own = timeit(lambda: kernels.conv2d_same(xin, w_hwio, None, groups=g))
    d = (kernels.conv2d_same(xin, w_hwio, None, groups=g) - F.conv2d(xin, w_oihw, None, 1, k//2, 1, g)).abs().max().item()
    print(f"{name}: MIOpen {t_mi:.1f} us ({gf/t_mi*1e-3*1e3/1e3:.1f} TF)  own {t_own:.1f} us ({gf/t_own:.4f} GF/us)  maxdiff {d:.2e}", flush=True)
for own in ((), ("conv2","conv3","conv4","conv5")):
    n2 = AlexNet(p, device=dev, own_conv=own)
    ms1 = timeit(lambda: n2.predict_records(rec[:B]), n=20)/1e3
    st = DeviceStage(n2, B, dev, n_streams=3)
    out = torch.empty((B*16, 6), device=dev)
    ms3 = timeit(lambda: st.run(rec, out), n=10, warm=2)/16/1e3
    print(f"own_conv={own}: eager {ms1:.3f} ms/batch; graph x3 streams {ms3:.3f} ms/batch", flush=True)
