#!/usr/bin/env python3
"""fc6 / fc7 at the CNN batch: svx_fc_bias_act (split count / tile shape forced through an experiment build) vs hipBLASLt."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
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


for m in (64, 128):
    for name, n, k in (("fc6", 4096, 9216), ("fc7", 4096, 4096)):
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        wp = kernels.pack_fc_weights(w)
        ws = torch.empty(64 * 1024 * 1024, device=dev)
        t_lt = timed(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
        row = ["%s m=%d hipBLASLt %.1f us" % (name, m, t_lt)]
        for na in (1, 2):
            for s in [int(v) for v in os.environ.get("SPLITS", "4,8,12,16,21,32").split(",")]:
                os.environ["SVX_FC_SPLITS"], os.environ["SVX_FC_NA"] = str(s), str(na)
                t = timed(lambda: kernels.fc_bias_act(x, wp, b, relu=True, ws=ws))
                row.append("na%d s%d %.1f" % (na, s, t))
        print(" | ".join(row), "  (weights %.0f MB: %.1f us at 6 TB/s)" % (n * k * 4 / 1e6, n * k * 4 / 6e6), flush=True)
