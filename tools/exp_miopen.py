#!/usr/bin/env python3
"""GPU experiment: MIOpen solver selection for conv3-5 (3x3, 13x13) at batch 64 fp32."""
import os, sys, time
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("TORCH_BENCH", "0") == "1"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
B=64
for name, cin, cout, g, hw, k, pad in (("conv2",96,256,2,27,5,2),("conv3",256,384,1,13,3,1),("conv4",384,384,2,13,3,1),("conv5",384,256,2,13,3,1)):
    for cl in (False, True):
        x = torch.randn(B,cin,hw,hw,device=dev); w = torch.randn(cout,cin//g,k,k,device=dev)*0.01
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        us = timeit(lambda: F.conv2d(x,w,None,1,pad,1,g))
        gf = 2*B*hw*hw*cout*(cin//g)*k*k/1e9
        print(f"{name} channels_last={cl}: {us:.1f} us  {gf/us*1e-3*1e3:.1f} TFLOP/s", flush=True)
