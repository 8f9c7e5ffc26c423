#!/usr/bin/env python3
"""GPU experiment: where does the per-window host time go?"""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights, options_ns
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import HotPath
from svision_amd.sample import Sample

dev = torch.device("cuda:0")
cfg = synth.SimConfig(contigs=[("chr21", 20_000_000)], coverage=30, seed=1)
table, genome, _ = synth.simulate(cfg)
opts = options_ns(64)
sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
net = AlexNet(random_weights(0), device=dev)
hot = HotPath(sample, opts, net, device=dev)
wins = [("chr21", 0, 10_000_000), ("chr21", 10_000_000, 20_000_000)]
for w in wins: hot.finish(hot.launch(hot.collect(*w)))
torch.cuda.synchronize()
def stage(name, fn):
    t = time.perf_counter(); r = fn(); dt = (time.perf_counter() - t) * 1e3
    print(f"{name:28s} {dt:8.2f} ms", flush=True); return r
for w in wins:
    stage("rescan_window", lambda: sample.rescan_window(*w))
    res = stage("collect (no rescan)", lambda: hot.collect(*w, rescan=False))
    stage("launch (async)", lambda: hot.launch(res))
    stage("sync (gpu wait)", lambda: torch.cuda.synchronize())
    stage("finish (d2h+vote)", lambda: hot.finish(res))
    print("images", res.n_images, "sites", res.n_sites, "records", res.n_records)
pr = cProfile.Profile(); pr.enable()
for w in wins: hot.finish(hot.launch(hot.collect(*w)))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
