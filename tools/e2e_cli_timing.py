#!/usr/bin/env python3
"""BAM-file-inclusive end-to-end timing of the CLI on a synthetic chr21-scale sample (for DESIGN.md section 6)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd import cli, synth
from svision_amd.io import bam
from svision_amd.network import tf_checkpoint as ck

length = int(sys.argv[1]) if len(sys.argv) > 1 else 46_709_983
d = tempfile.mkdtemp()
cfg = synth.SimConfig(contigs=[("chr21", length)], coverage=30, seed=1)
t = time.time(); table, genome, _ = synth.simulate(cfg); print("simulate %.1fs, %d records" % (time.time() - t, len(table)), flush=True)
t = time.time(); bam.write_bam(os.path.join(d, "s.bam"), table); print("write bam %.1fs, %.1f MB" % (time.time() - t, os.path.getsize(os.path.join(d, "s.bam")) / 1e6), flush=True)
bam.write_fasta(os.path.join(d, "g.fa"), genome)
ck.write_checkpoint(os.path.join(d, "m.ckpt"), random_weights(0))
opts = cli.parse_arguments(["-o", os.path.join(d, "out"), "-b", os.path.join(d, "s.bam"), "-m", os.path.join(d, "m.ckpt"),
                            "-g", os.path.join(d, "g.fa"), "-n", "S", "--batch_size", "64"])
t = time.time()
try:
    cli.run(opts)
except SystemExit as e:
    print("exit", e)
print("cli.run total %.2fs" % (time.time() - t))
for line in open([os.path.join(d, "out", f) for f in os.listdir(os.path.join(d, "out")) if f.endswith(".log")][0]):
    if "Cost time" in line or "finished" in line: print(line.rstrip())
