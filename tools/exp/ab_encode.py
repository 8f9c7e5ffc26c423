#!/usr/bin/env python3
"""encode_conv1 on the records of real candidate sites (128 per launch): time per launch and a checksum of the outputs
(SVX_EXP_LIB selects the build)."""
import os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["REAL"] = "1"
sys.argv = [sys.argv[0], "3"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prof_cnn.py")).read())
from svision_amd import kernels
x, t = kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)
e1.record(); torch.cuda.synchronize()
print("encode_conv1 %.1f us per launch of %d images; sha1 x %s touched %s" % (e0.elapsed_time(e1) / 50 * 1e3, rec.shape[0],
      hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12], hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]))
