"""BGZF inflate scaling of svx_bam_open on the box's host cores: seconds for a synthetic HiFi-like BAM vs threads."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from svision_amd import synth
from svision_amd.io import bam

path = "/tmp/scal.bam"
if not os.path.exists(path):
    table, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c%d" % i, 10_000_000) for i in range(4)], coverage=30.0, seed=2), with_genome=False)
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seed=t) for t in range(4)]
    bam.write_bam_segments(path, table.references, table.lengths, segs)
print("file MB", os.path.getsize(path) / 1e6, flush=True)
for zl in ("", "1"):
    for t in (1, 4, 16, 32, 64, 128):
        env = dict(os.environ, SVX_TIMING="1")
        if zl:
            env["SVX_BAM_ZLIB"] = "1"
        code = "import sys,time; sys.path.insert(0,%r); from svision_amd.io import bam; t=time.time(); x=bam.read_bam(%r, threads=%d); print('wall %%.3f' %% (time.time()-t))" % (os.path.join(os.path.dirname(__file__), "..", ".."), path, t)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(t, "zlib" if zl else "libdeflate", r.stdout.strip(), r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "", flush=True)
print("--- svx_bam_stream alone (parts dropped as they arrive)", flush=True)
for t in (16, 32, 48, 64, 96, 128, 192):
    code = ("import sys,time; sys.path.insert(0,%r); from svision_amd.io import bam; t=time.time(); n=sum(len(x) for x in bam.BamStream(%r, threads=%d)); "
            "print('wall %%.3f records %%d' %% (time.time()-t, n))" % (os.path.join(os.path.dirname(__file__), "..", ".."), path, t))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVX_TIMING="1"), capture_output=True, text=True)
    print(t, r.stdout.strip(), r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "", flush=True)
