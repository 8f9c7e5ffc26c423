#!/usr/bin/env python3
"""A few launches of the own implicit-GEMM conv on the AlexNet layer shapes (for rocprofv3 --pmc)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, cin, cout, g, hw, k in (("conv2",96,256,2,27,5),("conv3",256,384,1,13,3),("conv4",384,384,2,13,3),("conv5",384,256,2,13,3)):
    x = kernels.to_c8(torch.randn(64, cin, hw, hw, device=dev).clamp_min(0))
    w = kernels.pack_conv_weights(torch.randn(k, k, cin // g, cout, device=dev) * 0.02)
    for _ in range(3):
        kernels.conv2d_same(x, w, None, groups=g)
torch.cuda.synchronize()
