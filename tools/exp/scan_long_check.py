import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels, _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
    _lib.ABI_VERSION = 200
from tests import datagen
from oracle import cbind
dev = torch.device("cuda:0")
cigar, off, ref_start = datagen.random_cigars(6000, seed=77, mean_ops=4000, long_gap_rate=0.002, lognormal_sigma=1.0)
res = kernels.cigar_scan(torch.from_numpy(cigar.view(np.int32)).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ref_start).to(dev), 50)
gaps, gap_off, stats = res.to_host()
o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
print("gap_off equal", np.array_equal(gap_off, o_off), "stats equal", np.array_equal(stats, o_stats))
g = np.frombuffer(gaps.tobytes(), np.int32).reshape(-1, 6); og = np.frombuffer(o_gaps.tobytes(), np.int32).reshape(-1, 6)
bad = np.flatnonzero((g != og).any(1))
print("records", len(g), "differing", bad.size)
n_ops = np.diff(off.astype(np.int64))
for i in bad[:6]:
    a = og[i, 0]
    print("rec", i, "got", g[i].tolist(), "want", og[i].tolist(), "n_ops of aln", int(n_ops[a]), "first rec of aln", int(gap_off[a]), "n gaps", int(gap_off[a + 1] - gap_off[a]))
