#!/usr/bin/env python3
"""CPU-only profile of the host collection of one 10 Mb window (oracle scan stands in for the device scan)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import options_ns
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from tests import helpers
cfg = synth.SimConfig(contigs=[("chr21", 10_000_000)], coverage=30, seed=1)
table, genome, _ = synth.simulate(cfg)
opts = options_ns(64)
sample = Sample.with_scan(table, bam.Fasta(sequences=genome), 50, helpers.oracle_scan(table, 50))
for rep in range(2):
    t = time.perf_counter()
    sigs, clusters = detect_window(opts, sample, "chr21", 0, 10_000_000)
    t1 = time.perf_counter()
    lines = collect_pair_lines(clusters, opts)
    t2 = time.perf_counter()
    print("detect %.1f ms, lines %.1f ms, %d alignments, %d sigs, %d clusters, %d lines" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, len(table), len(sigs), len(clusters), len(lines)))
pr = cProfile.Profile(); pr.enable()
sigs, clusters = detect_window(opts, sample, "chr21", 0, 10_000_000)
lines = collect_pair_lines(clusters, opts)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
