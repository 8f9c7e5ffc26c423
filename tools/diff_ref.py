#!/usr/bin/env python3
"""Differential fuzzing of the host collection against the REFERENCE itself (this container only: imports
/root/reference through tests/golden/refdriver.py).  Random small samples x random option sets; the reference's
run_detect TSV is compared byte for byte with the product's (CPU path: oracle scan instead of the device scan).
    python tools/diff_ref.py [first_seed] [n_cases]
"""
import os, sys, shutil, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import refdriver
refdriver.install_stubs()
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.run_collection import detect_window
from svision_amd.collection.output_clusters import collect_pair_lines
from tests import helpers
from src.collection import run_collection as ref_run

MIXES = [
    (("DEL", 0.45), ("INS", 0.40), ("INV", 0.05), ("DUP", 0.05), ("dDUP", 0.03), ("DELINV", 0.02)),
    (("DEL", 0.2), ("INS", 0.2), ("INV", 0.2), ("DUP", 0.2), ("dDUP", 0.1), ("DELINV", 0.1)),
    (("INV", 0.3), ("DUP", 0.3), ("dDUP", 0.2), ("DELINV", 0.2)),
    (("DEL", 0.5), ("INS", 0.5)),
    (("cINS", 0.3), ("rcINS", 0.3), ("DUP", 0.2), ("INV", 0.2)),
]


def one_case(seed):
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.integers(1, 4))
    contigs = [("c%d" % i, int(rng.integers(120_000, 420_000))) for i in range(n_contigs)]
    lognormal = bool(rng.random() < 0.3)
    cfg = synth.SimConfig(contigs=contigs, coverage=float(rng.choice([6, 10, 16, 24])),
                          read_len_mean=float(rng.choice([3000, 6000, 9000, 14000])), read_len_sd=float(rng.choice([500, 1500, 3000])),
                          lognormal=lognormal, err_rate=float(rng.choice([0.0, 0.002, 0.01, 0.04])),
                          sv_spacing=float(rng.choice([3000, 6000, 12000])), sv_min_gap=int(rng.choice([2000, 5000, 9000])),
                          sv_max=int(rng.choice([800, 4000, 20000])), inline_max=int(rng.choice([300, 1500, 4000])),
                          het_frac=float(rng.choice([0.0, 0.5, 1.0])), seed=int(seed), sv_mix=MIXES[int(rng.integers(0, len(MIXES)))])
    use_hash = bool(rng.random() < 0.2)
    use_graph = bool(rng.random() < float(os.environ.get("GRAPH_FRAC", "0.25")))      # --graph: per-read .gfa trees compared as well
    table, genome, _svs = synth.simulate(cfg, with_seq=use_hash or use_graph)
    if os.environ.get("DUP_RECORDS") and not use_hash:
        # duplicate some records verbatim (value-equal segments of one read: analyze_reads.py:225 compares dicts by value)
        rows = np.arange(len(table))
        pick = rng.random(len(table)) < float(os.environ["DUP_RECORDS"])
        if os.environ.get("DUP_ONLY_SUPP"):
            pick &= (table.flag & 0x800) != 0
        table = table.subset(np.sort(np.concatenate([rows, rows[pick]]), kind="stable"))
    over = dict(hash=use_hash, graph=use_graph, min_support=int(rng.choice([1, 2, 3, 5, 8])), min_mapq=int(rng.choice([0, 10, 20, 40])),
                min_sv_size=int(rng.choice([30, 50, 100])), max_sv_size=int(rng.choice([3000, 1000000])),
                patition_max_distance=int(rng.choice([500, 5000])), cluster_max_distance=float(rng.choice([0.1, 0.3, 0.6])),
                contig=bool(rng.random() < 0.15), qname=bool(rng.random() < 0.2))
    window = int(rng.choice([60_000, 150_000, 10_000_000]))
    if over["contig"]:
        over["min_support"] = 1
    out = tempfile.mkdtemp()
    try:
        genome_path = os.path.join(out, "genome.fa")
        bam.write_fasta(genome_path, genome)
        refdriver.FASTAS.clear(); refdriver.DATASETS.clear()
        refdriver.FASTAS[genome_path] = genome
        refdriver.DATASETS["sample.bam"] = table
        os.mkdir(os.path.join(out, "segments"))
        gdir = os.path.join(out, "graphs")

        def graph_tree():
            tree = {}
            if os.path.exists(gdir):
                for d, _dirs, files in os.walk(gdir):
                    for name in files:
                        with open(os.path.join(d, name)) as f:
                            tree[os.path.relpath(os.path.join(d, name), gdir)] = f.read()
                shutil.rmtree(gdir)
            os.mkdir(gdir)
            return tree
        graph_tree()
        ropts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", window_size=window, **over)
        popts = helpers.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", window_size=window, **over)
        fasta = bam.Fasta(sequences=genome)
        scan = helpers.oracle_scan(table, over["min_sv_size"])
        n_lines = 0
        for chrom, clen in contigs:
            part, pos = 0, 0
            while pos < clen:
                end = clen if over["contig"] else min(clen, pos + window)
                err = ref_run.run_detect(ropts, "sample.bam", chrom, part, pos, end)
                path = os.path.join(out, "segments", "%s.segments.%d.bed" % (chrom, part))
                want = open(path).read() if os.path.exists(path) else None
                if os.path.exists(path):
                    os.remove(path)
                want_graphs = graph_tree()
                sample = Sample.with_scan(table, fasta, over["min_sv_size"], scan)
                try:                                         # run_detect's catch-all: a failing window writes nothing
                    _sigs, clusters = detect_window(popts, sample, chrom, pos, end, part)
                    got = "".join(p.text() for p in collect_pair_lines(clusters, popts))
                except Exception as exc:
                    got = ""
                    if err is None:
                        return "PRODUCT-ONLY EXCEPTION seed %d %s:%d-%d: %r" % (seed, chrom, pos, end, exc), n_lines
                got_graphs = graph_tree()
                if use_graph and err is None and want_graphs != got_graphs:
                    diff = sorted(k for k in set(want_graphs) | set(got_graphs) if want_graphs.get(k) != got_graphs.get(k))
                    return "GRAPH MISMATCH seed %d %s:%d-%d opts %s: %d of %d files differ, first %s" % (
                        seed, chrom, pos, end, over, len(diff), len(want_graphs), diff[:3]), n_lines
                if (want or "") != got:
                    return "MISMATCH seed %d %s:%d-%d part %d opts %s cfg %s (ref err %r): want %d lines, got %d" % (
                        seed, chrom, pos, end, part, over, cfg, err, (want or "").count("\n"), got.count("\n")), n_lines
                n_lines += got.count("\n")
                part += 1
                pos = end
        return None, n_lines
    finally:
        shutil.rmtree(out)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = total = 0
    for seed in range(first, first + n):
        try:
            msg, lines = one_case(seed)
        except Exception:
            msg, lines = "EXCEPTION seed %d\n%s" % (seed, traceback.format_exc()), 0
        total += lines
        if msg:
            bad += 1
            print(msg, flush=True)
    print("%d cases, %d mismatching, %d TSV lines compared" % (n, bad, total))
