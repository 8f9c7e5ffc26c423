#!/usr/bin/env python3
"""Does memory-bound work ride along with MFMA-bound work inside ONE launch?  List-mode conv2 with and without its
background-copy workgroups (61 % of the pixels copied from the background tensor), and the copy alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for nimg in (64, 256):
    cin, cout, g, hw, k, frac = 96, 256, 2, 27, 5, 0.39
    x = kernels.to_c8(torch.randn(nimg, cin, hw, hw, device=dev).clamp_min(0))
    w = kernels.pack_conv_weights(torch.randn(k, k, cin // g, cout, device=dev) * 0.02)
    npix = nimg * hw * hw
    act = torch.rand(npix, device=dev) < frac
    ids = torch.arange(npix, device=dev, dtype=torch.int32)
    plist = torch.cat([ids[act], ids[~act]]).contiguous()
    cnt = act.sum().to(torch.int32).view(1)
    zero = torch.zeros(1, dtype=torch.int32, device=dev)
    bg8 = kernels.to_c8(torch.randn(1, cout, hw, hw, device=dev))[0]
    out = torch.empty((nimg, cout // 8, hw, hw, 8), device=dev)
    t_both = timed(lambda: kernels.conv2d_same(x, w, None, groups=g, pixels=plist, pixel_count=cnt, background=bg8))
    t_conv = timed(lambda: kernels.conv2d_same(x, w, None, groups=g, pixels=plist, pixel_count=cnt, out=out))
    t_fill = timed(lambda: kernels.conv2d_same(x, w, None, groups=g, pixels=plist, pixel_count=zero, background=bg8))
    print("n=%d: compute + background copy %.1f us | compute only %.1f us | copy of ALL pixels only %.1f us (x %.2f = the inactive share)" % (nimg, t_both, t_conv, t_fill, 1 - frac))
