import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svision_amd import synth, _lib
from svision_amd.io import bam
cfg = synth.SimConfig(contigs=[("chr21", 46_709_983)], coverage=30, seed=1)
table, genome, _ = synth.simulate(cfg, with_genome=False)
d = tempfile.mkdtemp(); p = d + "/x.bam"
bam.write_bam(p, table)
print("file %.1f MB, %d records" % (os.path.getsize(p) / 1e6, len(table)))
lib = _lib.load()
for thr in (0, 8, 16, 32, 64):
    t = time.time(); h = lib.svx_bam_open(p.encode(), thr, 0); t1 = time.time() - t; lib.svx_bam_close(h)
    print("svx_bam_open threads=%d: %.3f s" % (thr, t1))
for ch in ("4000000", "16777216", "67108864"):
    os.environ["SVX_BAM_CHUNK"] = ch
    t = time.time(); h = lib.svx_bam_open(p.encode(), 16, 0); t1 = time.time() - t; lib.svx_bam_close(h)
    print("chunk %s threads 16: %.3f s" % (ch, t1))
del os.environ["SVX_BAM_CHUNK"]
t = time.time(); x = bam.read_bam(p); print("read_bam total %.3f s" % (time.time() - t))
