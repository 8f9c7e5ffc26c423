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
n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
threads = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1]
d = tempfile.mkdtemp()
cfg = synth.SimConfig(contigs=[("chr%d" % (21 + i), length) for i in range(n_contigs)], coverage=30, seed=1)
t = time.time(); table, genome, _ = synth.simulate(cfg); print("simulate %.1fs, %d records" % (time.time() - t, len(table)), flush=True)
t = time.time(); bam.write_bam(os.path.join(d, "s.bam"), table, index=True); print("write bam %.1fs, %.1f MB" % (time.time() - t, os.path.getsize(os.path.join(d, "s.bam")) / 1e6), flush=True)
bam.write_fasta(os.path.join(d, "g.fa"), genome)
ck.write_checkpoint(os.path.join(d, "m.ckpt"), random_weights(0))
import subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for nt in threads:                                   # a fresh process per run: the helpers must be forked before the first HIP call
    out = os.path.join(d, "out%d" % nt)
    code = ("import sys, time; sys.path.insert(0, %r); import torch; from svision_amd import cli; t = time.time(); "
            "m = cli.run(cli.parse_arguments(['-o', %r, '-b', %r, '-m', %r, '-g', %r, '-n', 'S', '--batch_size', '64', '-t', '%d'])); "
            "print('cli.run %%.2f s, %%d VCF records' %% (time.time() - t, sum(1 for l in open(m) if not l.startswith('#'))))"
            % (root, out, os.path.join(d, "s.bam"), os.path.join(d, "m.ckpt"), os.path.join(d, "g.fa"), nt))
    t = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SVX_TIMING="1"))
    print("-t %d: %s (process wall %.2f s)" % (nt, " | ".join(l for l in r.stdout.strip().splitlines() if "window " not in l and "owner: " not in l and "  helper " not in l) or r.stderr[-500:], time.time() - t), flush=True)
    for line in open([os.path.join(out, f) for f in os.listdir(out) if f.endswith(".log")][0]):
        if "Cost time" in line: print("   ", line.rstrip())
vcfs = {nt: open(os.path.join(d, "out%d" % nt, "S.svision.s5.vcf")).read() for nt in threads}
print("VCFs of -t %s identical: %s (%d records)" % (",".join(map(str, threads)), len(set(vcfs.values())) == 1, sum(1 for l in vcfs[threads[0]].splitlines() if not l.startswith("#"))))
