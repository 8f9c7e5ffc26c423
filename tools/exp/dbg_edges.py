import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels
from oracle import cbind
src = open("tests/test_gpu_kernels.py").read()
a = src.index("    rng = np.random.default_rng(5)\n    aligns = []"); b = src.index("    res = kernels.cigar_scan(_dev(padded.view(np.int32))")
exec("\n".join(l[4:] for l in src[a:b].splitlines()))
dev = torch.device("cuda:0")
_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for mode in ("groups4", "groups8", "groups8s"):
    res = kernels.cigar_scan(_dev(padded.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode, n_words=int(off[-1]))
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    print(mode, "total", int(gap_off[-1]), int(o_off[-1]), "off eq", np.array_equal(gap_off, o_off), "stats eq", np.array_equal(stats, o_stats))
    g = gaps.view(np.int32).reshape(-1, 6); o = o_gaps.view(np.int32).reshape(-1, 6)
    bad = np.flatnonzero((g != o).any(axis=1))
    print(" differing gaps", bad.size, "of", len(g))
    for i in bad[:12]:
        al = int(o[i, 0]); n = int(off[al + 1] - off[al]); b0 = int(off[al])
        print("  gap", i, "aln", al, "n", n, "b%4", b0 % 4, "got", g[i].tolist(), "want", o[i].tolist())
