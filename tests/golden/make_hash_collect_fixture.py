#!/usr/bin/env python3
"""Generates tests/golden/hash_collect.* : the REFERENCE's run_detect with --hash
(/root/reference/src/collection/run_collection.py:15, analyze_reads.py:731-790,898-929) on a small
synthetic sample whose primary records carry real read bases.  This container only."""
import gzip
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdriver  # noqa: E402

refdriver.install_stubs()
from svision_amd import synth  # noqa: E402
from svision_amd.io import bam  # noqa: E402

from src.collection import run_collection as ref_run  # noqa: E402  (reference)
from src.collection.collect_signatures import analyze_alignments as ref_analyze  # noqa: E402


def main():
    cfg = synth.SimConfig(contigs=[("chrH", 160_000)], coverage=10, read_len_mean=6000, read_len_sd=900, err_rate=0.003,
                          sv_spacing=3_000, sv_min_gap=5_000, sv_min=60, sv_max=900, inline_max=1000, seed=21,
                          sv_mix=(("cINS", 0.3), ("rcINS", 0.2), ("INS", 0.15), ("DEL", 0.1), ("DUP", 0.1), ("INV", 0.1), ("dDUP", 0.05)))
    table, genome, svs = synth.simulate(cfg, with_seq=True)
    bam_path = os.path.join(HERE, "hash_collect.bam")
    bam.write_bam(bam_path, table, level=9)
    with open(os.path.join(HERE, "hash_collect.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in genome.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    out = tempfile.mkdtemp()
    os.mkdir(os.path.join(out, "segments"))
    genome_path = os.path.join(out, "genome.fa")
    bam.write_fasta(genome_path, genome)
    refdriver.DATASETS["sample.bam"] = bam.read_bam(bam_path, with_seq=True)
    refdriver.FASTAS[genome_path] = genome
    expected = {"windows": []}
    for hash_on in (True, False):
        opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", min_support=3, hash=hash_on)
        import pysam
        aln = pysam.AlignmentFile("sample.bam")
        sigs = ref_analyze(aln.fetch("chrH", 0, 160_000), aln, opts, 0)
        dump = [[s.type, s.tstart, s.tend, s.qname, s.bkps, s.mechanism,
                 [[a["q_start"], a["q_end"], a["ref_start"], a["ref_end"], bool(a["is_reverse"])] for a in s.sorted_aligns]] for s in sigs]
        err = ref_run.run_detect(opts, "sample.bam", "chrH", 0, 0, 160_000)
        assert err is None, err
        tsv = open(os.path.join(out, "segments", "chrH.segments.0.bed")).read()
        expected["windows"].append({"hash": hash_on, "signatures": dump, "tsv": tsv})
        print("hash", hash_on, "signatures", len(dump), "with helpers", sum(1 for d in dump if len(d[6]) > 2), "tsv lines", tsv.count("\n"))
    shutil.rmtree(out)
    with open(os.path.join(HERE, "hash_collect.expected.json"), "w") as f:
        json.dump(expected, f)


if __name__ == "__main__":
    main()
