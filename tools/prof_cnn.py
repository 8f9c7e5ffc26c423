#!/usr/bin/env python3
"""Run N eager launches of the device stage (for rocprofv3 --kernel-trace --stats); IMAGES = images per launch
(default 256 = the pipeline's four batches of 64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd import _lib
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
from svision_amd.network.alexnet import AlexNet
from tests import datagen
dev = torch.device("cuda:0")
IMAGES = int(os.environ.get("IMAGES", "256"))
net = AlexNet(random_weights(0), device=dev)
if os.environ.get("REAL"):                        # records of real candidate sites (bench-like sample) instead of random segments
    from bench import options_ns
    from svision_amd import synth
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", 4_000_000)], coverage=30, seed=1))
    sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
    _s, clusters = detect_window(options_ns(64), sample, "chr21", 0, 4_000_000)
    lines = collect_pair_lines(clusters, options_ns(64))
    rec = torch.from_numpy(np.asarray([ln.record() for ln in lines[64:64 + IMAGES]], np.int32)).to(dev)
else:
    rec = torch.from_numpy(datagen.random_records(IMAGES, seed=1, hostile=False)).to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(3):
    net.predict_records(rec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    net.predict_records(rec)
e1.record()
torch.cuda.synchronize()
print("eager device stage: %.1f us per launch of %d images = %.1f us per batch of 64 (%s)" % (e0.elapsed_time(e1) / reps * 1e3, IMAGES, e0.elapsed_time(e1) / reps * 1e3 * 64 / IMAGES, os.environ.get("SVX_EXP_LIB")))
