#!/usr/bin/env python3
"""Generates tests/golden/collect_small.* by running the REFERENCE's run_detect
(/root/reference/src/collection/run_collection.py:15) on a small synthetic sample.
Run in this container only:  python tests/golden/make_collect_fixture.py
Outputs (data only): collect_small.bam, collect_small.fa.gz, collect_small.expected.json
"""
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
from src.collection.cluster_signatures import partition_and_cluster as ref_cluster  # noqa: E402


def main():
    cfg = synth.SimConfig(contigs=[("chrA", 420_000), ("chrB", 200_000)], coverage=16, read_len_mean=9000, read_len_sd=1500,
                          err_rate=0.004, sv_spacing=6_000, sv_min_gap=9_000, sv_max=4000, inline_max=1500, seed=11,
                          sv_mix=(("DEL", 0.3), ("INS", 0.3), ("INV", 0.12), ("DUP", 0.12), ("dDUP", 0.08), ("DELINV", 0.08)))
    table, genome, svs = synth.simulate(cfg)
    bam_path = os.path.join(HERE, "collect_small.bam")
    bam.write_bam(bam_path, table, level=9)
    with open(os.path.join(HERE, "collect_small.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in genome.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    refdriver.DATASETS["sample.bam"] = bam.read_bam(bam_path)
    out = tempfile.mkdtemp()
    genome_path = os.path.join(out, "genome.fa")       # run_detect open()s the path (run_collection.py:20)
    bam.write_fasta(genome_path, genome)
    refdriver.FASTAS[genome_path] = genome
    os.mkdir(os.path.join(out, "segments"))
    expected = {"windows": [], "n_records": len(table), "svs": svs}
    for min_support, window in ((3, 150_000), (5, 10_000_000)):
        opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", min_support=min_support,
                                         window_size=window)
        for chrom, clen in cfg.contigs:
            part, pos = 0, 0
            while pos < clen:
                end = min(clen, pos + window)
                # signatures and clusters (for fine-grained comparison), then the reference's own file output
                import pysam
                aln = pysam.AlignmentFile("sample.bam")
                sigs = ref_analyze(aln.fetch(chrom, pos, end), aln, opts, part)
                sig_dump = [[s.type, s.tstart, s.tend, s.qname, s.bkps, s.mechanism,
                             [[a["q_start"], a["q_end"], a["ref_start"], a["ref_end"], bool(a["is_reverse"])] for a in s.sorted_aligns]]
                            for s in sigs]
                clusters = ref_cluster(sigs, chrom, "sample.bam", opts)
                cl_dump = [[c.contig, c.cstart, c.cend, c.read_num, c.coverage, [s.qname for s in c.signatures]] for c in clusters]
                err = ref_run.run_detect(opts, "sample.bam", chrom, part, pos, end)
                assert err is None, err
                with open(os.path.join(out, "segments", "%s.segments.%d.bed" % (chrom, part))) as f:
                    tsv = f.read()
                expected["windows"].append({"min_support": min_support, "chrom": chrom, "part": part, "start": pos, "end": end,
                                            "signatures": sig_dump, "clusters": cl_dump, "tsv": tsv})
                part += 1
                pos = end
    shutil.rmtree(out)
    with open(os.path.join(HERE, "collect_small.expected.json"), "w") as f:
        json.dump(expected, f)
    n_sig = sum(len(w["signatures"]) for w in expected["windows"])
    n_lines = sum(w["tsv"].count("\n") for w in expected["windows"])
    print("records", len(table), "signatures", n_sig, "tsv lines", n_lines)
    from collections import Counter
    print(Counter(s[0] for w in expected["windows"] for s in w["signatures"]))


if __name__ == "__main__":
    main()
