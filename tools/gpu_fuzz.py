#!/usr/bin/env python3
"""GPU-side fuzzing (run on the MI355X box): for many simulated samples, (1) the resident device scan equals the oracle's,
(2) the TSV built on the device scan equals the one built on the oracle scan, (3) the active-set CNN path equals the dense
path bit for bit on the records of real candidate sites, (4) the packed softmax stays within 1e-3 of PyTorch-CPU fp32.
    python tools/gpu_fuzz.py [first_seed] [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.network.alexnet import AlexNet
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from tests import helpers
MIXES = [
    (("DEL", 0.45), ("INS", 0.40), ("INV", 0.05), ("DUP", 0.05), ("dDUP", 0.03), ("DELINV", 0.02)),
    (("DEL", 0.2), ("INS", 0.2), ("INV", 0.2), ("DUP", 0.2), ("dDUP", 0.1), ("DELINV", 0.1)),
    (("INV", 0.3), ("DUP", 0.3), ("dDUP", 0.2), ("DELINV", 0.2)),
    (("cINS", 0.3), ("rcINS", 0.3), ("DUP", 0.2), ("INV", 0.2)),
]
from oracle import cbind
dev = torch.device("cuda:0")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
params = random_weights(5)
from oracle.alexnet_torch import TorchAlexNet
dense, sparse, cpu = AlexNet(params, device=dev, active=False), AlexNet(params, device=dev, active=True), TorchAlexNet(params, device="cpu")
bad = n_rec = 0
worst = 0.0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    cfg = synth.SimConfig(contigs=[("c0", int(rng.integers(150_000, 500_000)))], coverage=float(rng.choice([8, 16, 30])),
                          read_len_mean=float(rng.choice([3000, 9000, 20000])), read_len_sd=2000.0, lognormal=bool(rng.random() < 0.4),
                          err_rate=float(rng.choice([0.0, 0.004, 0.05])), sv_spacing=float(rng.choice([3000, 8000])),
                          sv_min_gap=int(rng.choice([2000, 6000])), sv_max=int(rng.choice([1000, 8000])), inline_max=int(rng.choice([300, 2000])),
                          seed=int(seed), sv_mix=MIXES[int(rng.integers(0, len(MIXES)))])
    table, genome, _ = synth.simulate(cfg)
    fasta = bam.Fasta(sequences=genome)
    min_sv = int(rng.choice([30, 50, 100]))
    d = Sample.from_table(table, fasta, min_sv, device=dev)
    r = Sample.with_scan(table, fasta, min_sv, helpers.oracle_scan(table, min_sv))
    ok = d.gaps.tobytes() == r.gaps.tobytes() and np.array_equal(d.gap_off, r.gap_off) and np.array_equal(d.stats, r.stats)
    opts = helpers.default_options(min_support=int(rng.choice([1, 2, 3])), min_sv_size=min_sv)
    la = collect_pair_lines(detect_window(opts, d, "c0", 0, cfg.contigs[0][1])[1], opts)
    lb = collect_pair_lines(detect_window(opts, Sample.with_scan(table, fasta, min_sv, helpers.oracle_scan(table, min_sv)), "c0", 0, cfg.contigs[0][1])[1], opts)
    ok = ok and "".join(p.text() for p in la) == "".join(p.text() for p in lb)
    if la:
        rec_np = np.asarray([ln.record() for ln in la], np.int32)[:256]
        rec = torch.from_numpy(rec_np).to(dev)
        a, b = dense.predict_records_packed(rec), sparse.predict_records_packed(rec)
        ok = ok and torch.equal(a, b)
        sub = rec_np[:32]
        img = torch.from_numpy(cbind.rasterize(sub, "NCHW"))
        _l, _c, prob = cpu.predict(img)
        diff = float(np.abs(b[:len(sub), :5].cpu().numpy() - prob.numpy()).max())
        worst = max(worst, diff)
        ok = ok and diff < 1e-3
        n_rec += len(rec_np)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, cfg, flush=True)
print("%d samples, %d mismatching, %d records through both CNN paths, worst softmax difference vs PyTorch-CPU %.2e" % (n, bad, n_rec, worst))
