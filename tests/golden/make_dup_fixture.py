#!/usr/bin/env python3
"""Generates tests/golden/dup_small.* : a sample whose BAM holds some records TWICE (verbatim copies, as a careless
merge produces them), run through the REFERENCE's run_detect.  Upstream segments are dicts compared by value
(analyze_reads.py:53 `base_seg == target_seg`, :102/:126 in trim_segs, :225 `align in help_aligns`), so a duplicate
record counts as "itself": e.g. the copy of a middle same-strand segment is not "covered by another segment" and stays
a main segment.  The product mirrors that with Seg.same_value.  Run in this container only."""
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

from src.collection import run_collection as ref_run  # noqa: E402  (reference)


def main():
    cfg = synth.SimConfig(contigs=[("chrD", 260_000)], coverage=12, read_len_mean=9000, read_len_sd=1500, lognormal=True, err_rate=0.008,
                          sv_spacing=5_000, sv_min_gap=6_000, sv_max=6000, inline_max=1500, het_frac=0.3, seed=5021,
                          sv_mix=(("INV", 0.3), ("DUP", 0.3), ("dDUP", 0.2), ("DELINV", 0.1), ("DEL", 0.05), ("INS", 0.05)))
    table, genome, _ = synth.simulate(cfg)
    rng = np.random.default_rng(77)
    rows = np.arange(len(table))
    supp = (table.flag & 0x800) != 0
    pick = (supp & (rng.random(len(table)) < 0.5)) | (rng.random(len(table)) < 0.08)
    table = table.subset(np.sort(np.concatenate([rows, rows[pick]]), kind="stable"))
    bam_path = os.path.join(HERE, "dup_small.bam")
    bam.write_bam(bam_path, table, level=9)
    with open(os.path.join(HERE, "dup_small.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in genome.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    refdriver.DATASETS["sample.bam"] = bam.read_bam(bam_path)
    out = tempfile.mkdtemp()
    genome_path = os.path.join(out, "genome.fa")
    bam.write_fasta(genome_path, genome)
    refdriver.FASTAS[genome_path] = genome
    os.mkdir(os.path.join(out, "segments"))
    expected = {"duplicated_records": int(pick.sum()), "duplicated_supplementary": int((pick & supp).sum()), "cases": []}
    for over in (dict(min_support=2, min_mapq=10, window_size=130_000), dict(min_support=1, min_mapq=0, window_size=10_000_000, contig=True)):
        opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", **over)
        case = {"options": over, "windows": []}
        for chrom, clen in cfg.contigs:
            part, pos = 0, 0
            while pos < clen:
                end = clen if over.get("contig") else min(clen, pos + over["window_size"])
                err = ref_run.run_detect(opts, "sample.bam", chrom, part, pos, end)
                assert err is None, err
                p = os.path.join(out, "segments", "%s.segments.%d.bed" % (chrom, part))
                case["windows"].append({"chrom": chrom, "part": part, "start": pos, "end": end, "tsv": open(p).read() if os.path.exists(p) else ""})
                if os.path.exists(p):
                    os.remove(p)
                part, pos = part + 1, end
        expected["cases"].append(case)
    shutil.rmtree(out)
    with open(os.path.join(HERE, "dup_small.expected.json"), "w") as f:
        json.dump(expected, f)
    print("records", len(table), "duplicated", expected["duplicated_records"], "tsv lines",
          [sum(w["tsv"].count("\n") for w in c["windows"]) for c in expected["cases"]])


if __name__ == "__main__":
    main()
