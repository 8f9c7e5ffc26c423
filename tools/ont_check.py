#!/usr/bin/env python3
"""ONT-like stress (BASELINE configs[3] shape, scaled down): lognormal ultra-long reads, 5 % error (~10^4 CIGAR ops per read),
many supplementary pieces.  Device scan + streamed windows; TSV checked against the oracle-scan path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import options_ns, random_weights
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import HotPath
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from tests import helpers
dev = torch.device("cuda:0")
cfg = synth.SimConfig(contigs=[("chr20", 12_000_000)], coverage=30, read_len_mean=50_000, lognormal=True, lognormal_sigma=0.9,
                      err_rate=0.05, sv_spacing=60_000, seed=5)
t = time.time(); table, genome, _ = synth.simulate(cfg); print("simulate %.1f s: %d alignments, %d CIGAR ops (%.0f per alignment, max %d)" % (
    time.time() - t, len(table), table.cigar.size, table.cigar.size / len(table), int(np.diff(table.cig_off).max())))
opts = options_ns(64)
fasta = bam.Fasta(sequences=genome)
t = time.time(); sample = Sample.from_table(table, fasta, 50, device=dev); torch.cuda.synchronize(); print("upload + scan %.3f s, %d long gaps" % (time.time() - t, len(sample.gaps)))
ref = Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50))
assert sample.gaps.tobytes() == ref.gaps.tobytes() and np.array_equal(sample.stats, ref.stats)
net = AlexNet(random_weights(0), device=dev)
hot = HotPath(sample, opts, net, n_streams=3)
wins = [("chr20", s, min(12_000_000, s + 4_000_000)) for s in range(0, 12_000_000, 4_000_000)]
t = time.time(); n_img = n_sites = 0
for res in hot.run_windows(wins):
    n_img += res.n_images; n_sites += res.n_sites
print("3 windows: %.2f s, %d images, %d sites" % (time.time() - t, n_img, n_sites))
for w in wins[:1]:
    a = "".join(p.text() for p in collect_pair_lines(detect_window(opts, sample, *w)[1], opts))
    b = "".join(p.text() for p in collect_pair_lines(detect_window(opts, ref, *w)[1], opts))
    assert a == b
print("ONT-like check ok")
