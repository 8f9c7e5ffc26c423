#!/usr/bin/env python3
"""CIGAR scan timing at two sizes (HIP events): 2 M alignments x ~147 ops (leaves the caches) and a
chr21-sized 100 k alignments (launch-latency regime)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
from tests import datagen
dev = torch.device("cuda:0")
reps = int(os.environ.get("REPS", "20"))
out = {}
SIZES = ((2_000_000, 150), (100_000, 150), (50_000, 6000), (35_000, -5000), (416_000, 145), (250_000, 300), (150_000, 500), (100_000, 800))      # negative: log-normal op counts (sigma 0.7) around the median
if os.environ.get("ONLY"):
    SIZES = tuple(SIZES[int(i)] for i in os.environ["ONLY"].split(","))
for na, mean_ops in SIZES:
    cigar, off, ref_start = datagen.random_cigars(na, seed=5, mean_ops=abs(mean_ops), long_gap_rate=0.0005, lognormal_sigma=0.7 if mean_ops < 0 else None)
    d_c = torch.from_numpy(cigar.view(np.int32)).to(dev); d_o = torch.from_numpy(off.astype(np.int64)).to(dev); d_r = torch.from_numpy(ref_start).to(dev)
    cap = 1 << 22
    res = kernels.cigar_scan(d_c, d_o, d_r, 50, gaps_cap=cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kernels.cigar_scan(d_c, d_o, d_r, 50, gaps_cap=cap)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    alg = 4 * cigar.size + 32 * na + 24 * res.total()
    n_ops = np.diff(off.astype(np.int64))
    out["%d x %d" % (na, mean_ops)] = {"us": t * 1e6, "GBps": alg / t / 1e9, "gaps": res.total(), "max_ops": int(n_ops.max())}
print(json.dumps(out))
