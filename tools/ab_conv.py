#!/usr/bin/env python3
"""Per-layer timing of svx_conv2d_same on the AlexNet shapes (batch 64), dense and list mode (HIP events).

SVX_EXP_LIB=<libsvx built with -DSVX_CONV_EXPERIMENT>: every wave-tile shape is forced in turn (SVX_CONV_SHAPE) and the
outputs are required to be bit-identical across shapes; without it only the library's own choice is timed."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels, _lib
EXP = os.environ.get("SVX_EXP_LIB")
if EXP:
    _lib.LIB_PATH = EXP
SHAPES = [None] + (list(range(int(os.environ.get("SVX_N_SHAPES", "5")))) if EXP and os.environ.get("SVX_FORCE", "1") == "1" else [])
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


tot = {}
for name, cin, cout, g, hw, k, fused, frac in (("conv2", 96, 256, 2, 27, 5, False, 0.37), ("conv3", 256, 384, 1, 13, 3, True, 0.67),
                                                ("conv4", 384, 384, 2, 13, 3, True, 0.81), ("conv5", 384, 256, 2, 13, 3, False, 0.91)):
    x_nchw = torch.randn(64, cin, hw, hw, device=dev).clamp_min(0)
    w_hwio = torch.randn(k, k, cin // g, cout, device=dev) * 0.02
    b = torch.randn(cout, device=dev) if fused else None
    ref = F.conv2d(x_nchw, w_hwio.permute(3, 2, 0, 1).contiguous(), b, 1, k // 2, 1, g)
    x, w = kernels.to_c8(x_nchw), kernels.pack_conv_weights(w_hwio)
    if fused:
        ref = ref.clamp_min(0)
    fl = 2.0 * 64 * hw * hw * cout * (cin // g) * k * k
    npix = 64 * hw * hw
    act = torch.rand(npix, device=dev) < frac                   # active pixels; list = active ascending, then inactive ascending
    ids = torch.arange(npix, device=dev, dtype=torch.int32)
    plist = torch.cat([ids[act], ids[~act]]).contiguous()
    cnt = act.sum().to(torch.int32).view(1)
    bg = torch.randn(cout, hw, hw, device=dev)
    bg8 = kernels.to_c8(bg.unsqueeze(0))[0]
    want_list = torch.where(act.view(64, 1, hw, hw), ref, bg.unsqueeze(0).expand(64, -1, -1, -1))
    for mode in ("dense", "list"):
        outs = {}
        for sh in SHAPES:
            if sh is None:
                os.environ.pop("SVX_CONV_SHAPE", None)
            else:
                os.environ["SVX_CONV_SHAPE"] = str(sh)
            if mode == "dense":
                fn = lambda: kernels.conv2d_same(x, w, b, groups=g, relu=fused)
                want, work = ref, fl
            else:
                fn = lambda: kernels.conv2d_same(x, w, b, groups=g, relu=fused, pixels=plist, pixel_count=cnt, background=bg8)
                want, work = want_list, fl * float(cnt.item()) / npix
            t = timed(fn)
            out = kernels.from_c8(fn())
            outs[sh] = out
            d = (out - want).abs().max().item()
            same = "" if sh is None else (" bit-identical to default" if torch.equal(out, outs[None]) else " DIFFERS from default")
            print("%s %-5s shape %-4s: %7.1f us  %6.1f TF executed (%.3f of 157.3)  maxdiff %.2e%s" % (name, mode, sh, t, work / t / 1e6, work / t / 1e6 / 157.3, d, same), flush=True)
            if sh is None:
                tot[mode] = tot.get(mode, 0) + t
print("sum dense %.1f us, list %.1f us (%s)" % (tot["dense"], tot["list"], EXP or "product lib"))
