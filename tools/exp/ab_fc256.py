#!/usr/bin/env python3
"""fc6 + fc7 at the pipeline's launch size (M = 256): time per launch (SVX_EXP_LIB selects the build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
for m in (64, 128, 256):
    x = torch.randn(m, 9216, device=dev)
    w6 = kernels.pack_fc_weights(torch.randn(4096, 9216, device=dev) * 0.01); b6 = torch.randn(4096, device=dev)
    w7 = kernels.pack_fc_weights(torch.randn(4096, 4096, device=dev) * 0.01)
    def go():
        return kernels.fc_bias_act(kernels.fc_bias_act(x, w6, b6, relu=True), w7, b6, relu=True)
    for _ in range(3): y = go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): go()
    e1.record(); torch.cuda.synchronize()
    print("M=%d fc6+fc7 %.1f us  checksum %.6f" % (m, e0.elapsed_time(e1) / 30 * 1e3, float(y.double().sum())))
