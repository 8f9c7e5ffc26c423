#!/usr/bin/env python3
"""Generates tests/golden/modes.expected.json (+ ont_small.bam / .fa.gz): the REFERENCE's run_detect in
  * --contig mode (min_mapq 0, no supplementary cap, min_support 1, whole-contig window; SVision:161-180,
    collect_signatures.py:125, analyze_reads.py:628-633) on collect_small.bam, and
  * an ONT-like sample (long log-normal reads, 4 % error, several SVs per read, > 4 supplementary
    alignments on some reads, low-MAPQ / secondary / unmapped records, 100 kb windows so that reads are
    seen by two windows).  This container only."""
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdriver  # noqa: E402

refdriver.install_stubs()
from svision_amd import synth  # noqa: E402
from svision_amd.io import bam  # noqa: E402
from tests import helpers  # noqa: E402

from src.collection import run_collection as ref_run  # noqa: E402  (reference)


def run_windows(tag, table, genome, windows, **opt):
    out = tempfile.mkdtemp()
    os.mkdir(os.path.join(out, "segments"))
    genome_path = os.path.join(out, "genome.fa")
    bam.write_fasta(genome_path, genome)
    refdriver.DATASETS["sample.bam"] = table
    refdriver.FASTAS[genome_path] = genome
    opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", **opt)
    res = []
    for part, (chrom, start, end) in enumerate(windows):
        err = ref_run.run_detect(opts, "sample.bam", chrom, part, start, end)
        assert err is None, err
        tsv = open(os.path.join(out, "segments", "%s.segments.%d.bed" % (chrom, part))).read()
        res.append({"chrom": chrom, "start": start, "end": end, "tsv": tsv})
        print(tag, chrom, start, end, "tsv lines", tsv.count("\n"))
    shutil.rmtree(out)
    return {"options": opt, "windows": res}


def main():
    expected = {}
    # --contig on the collect_small sample
    table = bam.read_bam(os.path.join(HERE, "collect_small.bam"))
    fasta = helpers.load_golden_fasta()
    genome = {n: fasta._seq[n] for n in fasta.references}
    expected["contig"] = run_windows("contig", table, genome, [("chrA", 0, 420_000), ("chrB", 0, 200_000)],
                                     contig=True, min_support=1)
    # ONT-like sample
    cfg = synth.SimConfig(contigs=[("chrO", 260_000)], coverage=9, read_len_mean=14_000, lognormal=True, lognormal_sigma=0.6,
                          err_rate=0.04, sv_spacing=1_500, sv_min_gap=2_500, sv_min=60, sv_max=2_500, inline_max=800, seed=31,
                          sv_mix=(("DEL", 0.25), ("INS", 0.25), ("INV", 0.2), ("DUP", 0.1), ("dDUP", 0.1), ("DELINV", 0.1)))
    t, g, _svs = synth.simulate(cfg)
    rng = np.random.default_rng(7)
    low = rng.random(len(t)) < 0.05
    t.mapq[low] = 3
    sec = rng.random(len(t)) < 0.02
    t.flag[sec] |= 0x100
    unm = rng.random(len(t)) < 0.01
    t.flag[unm] |= 0x4
    bam.write_bam(os.path.join(HERE, "ont_small.bam"), t, level=9)
    with open(os.path.join(HERE, "ont_small.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in g.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    t = bam.read_bam(os.path.join(HERE, "ont_small.bam"))
    n_ops = t.cig_off[1:] - t.cig_off[:-1]
    supp = np.bincount(t.name_id[(t.flag & 0x800) != 0], minlength=len(t.names))
    print("ont_small: records", len(t), "mean ops", float(n_ops.mean()), "max ops", int(n_ops.max()), "reads with >4 supp", int((supp > 4).sum()))
    expected["ont"] = run_windows("ont", t, g, [("chrO", 0, 100_000), ("chrO", 100_000, 200_000), ("chrO", 200_000, 260_000)],
                                  min_support=2)
    with open(os.path.join(HERE, "modes.expected.json"), "w") as f:
        json.dump(expected, f)


if __name__ == "__main__":
    main()
