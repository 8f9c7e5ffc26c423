#!/usr/bin/env python3
"""Per-layer timing of svx_conv2d_same on the AlexNet shapes (batch 64), with and without bias (HIP events)."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0
for name, cin, cout, g, hw, k, fused in (("conv2",96,256,2,27,5,False),("conv3",256,384,1,13,3,True),("conv4",384,384,2,13,3,True),("conv5",384,256,2,13,3,False)):
    x = torch.randn(64, cin, hw, hw, device=dev).clamp_min(0)
    w = torch.randn(k, k, cin // g, cout, device=dev) * 0.02
    b = torch.randn(cout, device=dev) if fused else None
    t = timed(lambda: kernels.conv2d_same(x, w, b, groups=g, relu=fused))
    ref = F.conv2d(x, w.permute(3, 2, 0, 1).contiguous(), b, 1, k // 2, 1, g)
    if fused: ref = ref.clamp_min(0)
    d = (kernels.conv2d_same(x, w, b, groups=g, relu=fused) - ref).abs().max().item()
    fl = 2.0 * 64 * hw * hw * cout * (cin // g) * k * k
    tot += t
    print("%s: %.1f us  %.1f TF  maxdiff %.2e" % (name, t, fl / t / 1e6, d))
print("sum %.1f us (%s)" % (tot, os.environ.get("SVX_EXP_LIB")))
