"""Host-side profile of one window's collection (CPU only: the scan comes from the C oracle)."""
import cProfile, pstats, sys, time, os, pickle
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import bench
from oracle import cbind
from svision_amd import synth
from svision_amd.sample import Sample
from svision_amd.io.bam import Fasta
from svision_amd.pipeline import _collect_lines

cache = "/tmp/prof_collect_workload.pkl"
if os.path.exists(cache):
    table, genome = pickle.load(open(cache, "rb"))
else:
    table, genome, _seg = bench._simulate_contig(dict(name="chr21", length=46709983, coverage=30, seed=1, kind=None))
    pickle.dump((table, genome), open(cache, "wb"))
opts = bench.options_ns(64)
fasta = Fasta(sequences={"chr21": genome})
scan = cbind.cigar_scan(table.cigar, table.cig_off.astype(np.uint64), table.pos, opts.min_sv_size)
sample = Sample.with_scan(table, fasta, opts.min_sv_size, scan)
wins = bench.windows_of("chr21", 46709983)
for w in wins[:2]:
    t = time.perf_counter(); lines = _collect_lines(sample, opts, *w); print(w, len(lines), "lines", round((time.perf_counter() - t) * 1e3, 1), "ms")
pr = cProfile.Profile(); pr.enable()
for w in wins[1:4]:
    lines = _collect_lines(sample, opts, *w)
    recs = np.asarray([ln.record() for ln in lines], np.int32).reshape(-1, 12)
pr.disable()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(40)
