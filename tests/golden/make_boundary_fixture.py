#!/usr/bin/env python3
"""Generates tests/golden/boundary_small.* : a sample whose SVs sit within a read length of the collection-window
boundaries, so that the same cluster -- same region string chrom+cstart+cend+coverage -- is the last site of window k
and the first site of window k+1 (reads overlapping the boundary are fetched by both windows, run_collection.py:26).
The REFERENCE votes once over the concatenated {chrom}.segments.all.bed (predict.py:235-247: a site ends when the
region string changes), so such a pair is ONE site and one VCF record upstream.  Expected outputs = the reference's
run_detect per window, then its Predict.run / write_results_to_vcf / merge_split_vcfs over the concatenation, with the
TensorFlow session replaced by the deterministic pseudo-classifier of make_predict_fixture (outputs stored).
Run in this container only."""
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
from make_predict_fixture import pseudo_classifier  # noqa: E402

from src.collection import run_collection as ref_run  # noqa: E402  (reference)
from src.network.predict import Predict as RefPredict  # noqa: E402
from src.network.output import cal_scores_max_min as ref_scores, merge_split_vcfs as ref_merge  # noqa: E402

WINDOW = 100_000


def main():
    cfg = synth.SimConfig(contigs=[("chrW", 410_000), ("chrV", 150_000)], coverage=22, read_len_mean=9000, read_len_sd=1500,
                          err_rate=0.003, inline_max=1500, seed=29)
    hom = (1, 1)
    svs = {"chrW": [{"type": "DEL", "pos": 30_000, "len": 400, "src": -1, "gt": hom},
                    {"type": "DEL", "pos": 99_100, "len": 300, "src": -1, "gt": hom},        # 900 bp before the first boundary
                    {"type": "INS", "pos": 140_000, "len": 250, "src": -1, "gt": (1, 0)},
                    {"type": "INS", "pos": 199_600, "len": 220, "src": -1, "gt": hom},       # 400 bp before the second boundary
                    {"type": "DEL", "pos": 300_400, "len": 500, "src": -1, "gt": hom},       # 400 bp behind the third boundary
                    {"type": "INV", "pos": 350_000, "len": 900, "src": -1, "gt": hom}],
           "chrV": [{"type": "DEL", "pos": 98_500, "len": 350, "src": -1, "gt": hom},
                    {"type": "DUP", "pos": 120_000, "len": 700, "src": -1, "gt": hom}]}
    table, genome, _ = synth.simulate(cfg, svs=svs)
    bam_path = os.path.join(HERE, "boundary_small.bam")
    bam.write_bam(bam_path, table, level=9)
    with open(os.path.join(HERE, "boundary_small.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in genome.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    refdriver.DATASETS["sample.bam"] = bam.read_bam(bam_path)
    out = tempfile.mkdtemp()
    genome_path = os.path.join(out, "genome.fa")
    bam.write_fasta(genome_path, genome)
    refdriver.FASTAS[genome_path] = genome
    seg_dir, pred_dir = os.path.join(out, "segments"), os.path.join(out, "predict_results")
    os.mkdir(seg_dir)
    os.mkdir(pred_dir)
    opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", min_support=4, window_size=WINDOW,
                                     batch_size=64, model_path="unused.ckpt", sample="HGb")
    expected = {"window": WINDOW, "min_support": 4, "batch_size": 64, "chroms": {}, "chrom_order": [c for c, _l in cfg.contigs]}
    shared = 0
    for chrom, clen in cfg.contigs:
        wins, part, pos = [], 0, 0
        while pos < clen:
            end = min(clen, pos + WINDOW)
            err = ref_run.run_detect(opts, "sample.bam", chrom, part, pos, end)
            assert err is None, err
            p = os.path.join(seg_dir, "%s.segments.%d.bed" % (chrom, part))
            wins.append({"start": pos, "end": end, "tsv": open(p).read() if os.path.exists(p) else ""})
            part, pos = part + 1, end
        bed = os.path.join(seg_dir, chrom + ".segments.all.bed")
        with open(bed, "w") as f:                                  # the driver's `cat` (SVision:284-288)
            f.write("".join(w["tsv"] for w in wins))
        for a, b in zip(wins, wins[1:]):
            if a["tsv"] and b["tsv"] and a["tsv"].splitlines()[-1].split("\t")[0] == b["tsv"].splitlines()[0].split("\t")[0]:
                shared += 1
        preds = []

        def fn(batch, preds=preds):
            lo, cl, pr = pseudo_classifier(batch)
            preds.append((cl.copy(), pr.copy()))
            return lo, cl, pr
        refdriver.PREDICTOR["fn"] = fn
        prefix = os.path.join(pred_dir, "%s.predict.s%d" % (chrom, opts.min_support))
        RefPredict(chrom, bed).run(prefix, opts)
        expected["chroms"][chrom] = {
            "windows": wins, "vcf": open(prefix + ".vcf").read(), "score": open(prefix + ".score.txt").read(),
            "classes": np.concatenate([p[0] for p in preds]).tolist(),
            "probs": np.concatenate([p[1] for p in preds]).astype(np.float32).view(np.uint32).tolist()}
    scores = ref_scores(pred_dir)
    mx, mn = np.max(scores), np.min(scores)
    merged = os.path.join(out, "merged.vcf")
    ref_merge(pred_dir, merged, mx, mn, expected["chrom_order"], opts)
    expected["merged_vcf"] = open(merged).read()
    expected["max_score"], expected["min_score"] = float(mx), float(mn)
    expected["boundary_sites"] = shared
    shutil.rmtree(out)
    assert shared >= 3, "no region string is shared by adjacent windows: move the SVs"
    with open(os.path.join(HERE, "boundary_small.expected.json"), "w") as f:
        json.dump(expected, f)
    print("records", len(table), "boundary sites", shared, "vcf records", expected["merged_vcf"].count("\n"),
          {c: [w["tsv"].count("\n") for w in v["windows"]] for c, v in expected["chroms"].items()})


if __name__ == "__main__":
    main()
