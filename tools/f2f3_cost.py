#!/usr/bin/env python3
"""Do the SURVEY 8(f)2 / 8(f)3 kernels pay inside ``-t N``?  

In ``-t N`` the collection of a window runs in a forked host helper that never touches the GPU; the clustering distance
matrices (cluster_signatures.py:110-115) and the ``--hash`` seed passes (run_hash_lineplot.py:64-78) would have to travel
to the GPU-owning process and back.  This measures, on the GPU box and on the workload's own data:

  f2  every partition of a bench window: NumPy condensed matrices in the helper (what -t N runs) vs
      svx_span_position_distance in one launch (upload + kernel + read-back, what -t 1 runs) vs the same plus a pipe round
      trip to another process (what routing it through the owner would cost a helper);
  f3  the hash_small fixture's (window, piece) jobs: the Python seed-and-extend passes vs svx_hash_seeds (one launch
      for all of them, and one launch per job as a helper meeting them one by one would issue them).

Prints one JSON object (committed as profiles/r03_f2f3_cost.json)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _echo(conn):
    while True:
        msg = conn.recv()
        if msg is None:
            return
        conn.send(msg)


def main():
    import torch
    ctx = mp.get_context("fork")
    a, b = ctx.Pipe(duplex=True)
    echo = ctx.Process(target=_echo, args=(b,), daemon=True)
    echo.start()                                              # before the first HIP call
    import bench
    from svision_amd import kernels, synth
    from svision_amd.collection import cluster_signatures as cs
    from svision_amd.collection.collect_signatures import analyze_alignments
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    dev = torch.device("cuda:0")
    out = {}

    # ---- f2: the partitions of real windows
    table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", 30_000_000)], coverage=30.0, seed=1))
    sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, dev)
    opts = bench.options_ns(64)
    rows = []
    for w in range(3):
        sigs = analyze_alignments(sample.table.fetch(0, w * 10_000_000, (w + 1) * 10_000_000), sample, opts, 0)
        parts = [p for p in cs.signature_partition(sigs, opts) if len(p) > 1]
        sizes = [len(p) for p in parts]
        reps = 20
        t = time.perf_counter()
        for _ in range(reps):
            host = [cs.span_position_distance_condensed([s.tstart for s in p], [s.tend for s in p]) for p in parts]
        t_numpy = (time.perf_counter() - t) / reps
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            got = cs.condensed_distances(parts, sample)                  # device: upload, one launch, read-back
        t_dev = (time.perf_counter() - t) / reps
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(host, got))
        payload = ([s.tstart for p in parts for s in p], [s.tend for p in parts for s in p], sizes)
        t = time.perf_counter()
        for _ in range(reps):
            a.send(payload)
            a.recv()
        t_pipe = (time.perf_counter() - t) / reps
        t = time.perf_counter()
        for _ in range(reps):
            from scipy.cluster.hierarchy import fcluster, linkage
            for d in host:
                fcluster(linkage(d, method="average"), 0.3, criterion="distance")
        t_linkage = (time.perf_counter() - t) / reps
        rows.append({"window": w, "signatures": len(sigs), "partitions": len(parts), "largest": max(sizes), "pairs": int(sum(n * (n - 1) // 2 for n in sizes)),
                     "numpy_ms": t_numpy * 1e3, "device_ms": t_dev * 1e3, "pipe_round_trip_ms": t_pipe * 1e3, "scipy_linkage_ms": t_linkage * 1e3})
    out["f2_span_position_distance"] = rows

    # ---- f3: the hash fixture's jobs
    from svision_amd.segmentplot import run_hash_lineplot as rh
    with open(os.path.join(ROOT, "tests", "golden", "hash_small.expected.json")) as f:
        cases = json.load(f)
    pairs = [(c["ref"], c["seq"]) for c in cases if 0 < len(c["seq"]) <= kernels.HASH_MAX_X]
    t = time.perf_counter()
    for ref, seq in pairs:
        rh._hashplot_host(ref, seq, 10, 50)
    t_host = time.perf_counter() - t
    rh.hashplot_unmapped_batch(pairs[:4], 10, 50, dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    rh.hashplot_unmapped_batch(pairs, 10, 50, dev)
    t_batch = time.perf_counter() - t
    t = time.perf_counter()
    for p in pairs:
        rh.hashplot_unmapped_batch([p], 10, 50, dev)
    t_single = time.perf_counter() - t
    out["f3_hash_seeds"] = {"jobs": len(pairs), "mean_window": float(np.mean([len(r) for r, _s in pairs])), "mean_piece": float(np.mean([len(s) for _r, s in pairs])),
                            "python_passes_ms_per_job": t_host / len(pairs) * 1e3, "device_one_launch_for_all_ms_per_job": t_batch / len(pairs) * 1e3,
                            "device_one_launch_per_job_ms_per_job": t_single / len(pairs) * 1e3}
    a.send(None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
