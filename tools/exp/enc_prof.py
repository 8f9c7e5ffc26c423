#!/usr/bin/env python3
"""Phase times inside encode_conv1_kernel (experiment build with -DSVX_ENC_PROFILE): wall-clock ticks (100 MHz) summed
over workgroups by thread 0 of each."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import _lib
_lib.LIB_PATH = os.environ.get("SVX_EXP_LIB", "tools/exp/libsvx_encprof.so")
os.environ["REAL"] = "1"
sys.argv = [sys.argv[0], "5"]
exec(open(os.path.join(os.path.dirname(__file__), "..", "prof_cnn.py")).read())
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
lib.svx_debug_enc_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.svx_debug_enc_prof(out, 1)
from svision_amd import kernels
for _ in range(10):
    kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)
torch.cuda.synchronize()
lib.svx_debug_enc_prof(out, 0)
v = list(out)
wgs = v[4]
print("workgroups %d, touched windows per workgroup %.1f" % (wgs, v[5] / wgs))
for name, t in zip(("draw_planes", "masks + queue", "tap accumulation", "lrn + store"), v[:4]):
    print("%-18s %.2f us per workgroup" % (name, t / wgs / 100.0))
