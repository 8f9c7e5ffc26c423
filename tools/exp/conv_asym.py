#!/usr/bin/env python3
"""Asymptotic efficiency of the wave-tile convolution: executed TFLOP/s of each AlexNet layer at 64 / 256 / 1024 images
(dense and list mode; at 1024 images the tile-count quantisation of one launch is negligible), and of 64-image launches
issued round-robin on 1 / 4 streams."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, reps=20):
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


LAYERS = (("conv2", 96, 256, 2, 27, 5, False, 0.39), ("conv3", 256, 384, 1, 13, 3, True, 0.67),
          ("conv4", 384, 384, 2, 13, 3, True, 0.81), ("conv5", 384, 256, 2, 13, 3, False, 0.91))
for name, cin, cout, g, hw, k, fused, frac in LAYERS:
    for nimg in (64, 256, 1024):
        x = kernels.to_c8(torch.randn(nimg, cin, hw, hw, device=dev).clamp_min(0))
        w = kernels.pack_conv_weights(torch.randn(k, k, cin // g, cout, device=dev) * 0.02)
        b = torch.randn(cout, device=dev) if fused else None
        fl = 2.0 * nimg * hw * hw * cout * (cin // g) * k * k
        npix = nimg * hw * hw
        act = torch.rand(npix, device=dev) < frac
        ids = torch.arange(npix, device=dev, dtype=torch.int32)
        plist = torch.cat([ids[act], ids[~act]]).contiguous()
        cnt = act.sum().to(torch.int32).view(1)
        bg8 = kernels.to_c8(torch.randn(1, cout, hw, hw, device=dev))[0]
        td = timed(lambda: kernels.conv2d_same(x, w, b, groups=g, relu=fused), reps=20 if nimg < 1024 else 5)
        tl = timed(lambda: kernels.conv2d_same(x, w, b, groups=g, relu=fused, pixels=plist, pixel_count=cnt, background=bg8), reps=20 if nimg < 1024 else 5)
        wl = fl * float(cnt.item()) / npix
        print("%s n=%4d  dense %8.1f us %6.1f TF (%.3f)   list %8.1f us %6.1f TF (%.3f)" % (name, nimg, td, fl / td / 1e6, fl / td / 1e6 / 157.3, tl, wl / tl / 1e6, wl / tl / 1e6 / 157.3), flush=True)
        if nimg == 64:
            streams = [torch.cuda.Stream() for _ in range(4)]
            outs = [None] * 4
            for mode in ("dense", "list"):
                for ns in (1, 4):
                    def go():
                        for i in range(8):
                            with torch.cuda.stream(streams[i % ns]):
                                if mode == "dense":
                                    kernels.conv2d_same(x, w, b, groups=g, relu=fused)
                                else:
                                    kernels.conv2d_same(x, w, b, groups=g, relu=fused, pixels=plist, pixel_count=cnt, background=bg8)
                    for _ in range(2):
                        go()
                    torch.cuda.synchronize()
                    import time
                    t = time.perf_counter()
                    for _ in range(10):
                        go()
                    torch.cuda.synchronize()
                    us = (time.perf_counter() - t) / 80 * 1e6
                    work = fl if mode == "dense" else wl
                    print("      %-5s x8 round-robin on %d stream(s): %7.1f us per launch  %6.1f TF (%.3f)" % (mode, ns, us, work / us / 1e6, work / us / 1e6 / 157.3), flush=True)
