#!/usr/bin/env python3
"""Differential fuzzing of the vote / VCF / genotype / merge stage against the REFERENCE itself (this container only).
Random small samples -> segment TSVs (product collection, itself fuzzed by tools/diff_ref.py) -> the reference's
Predict.run + merge_split_vcfs with a random classifier vs the product's with the same classifier outputs injected.
    python tools/diff_ref_predict.py [first_seed] [n_cases]
"""
import os, sys, shutil, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch  # noqa: F401  (before the stub modules are installed)
import refdriver
refdriver.install_stubs()
from svision_amd import synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.run_collection import detect_window
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.network import output
from svision_amd.network.predict import Predict
from tests import helpers
from tools.diff_ref import MIXES
from src.network.predict import Predict as RefPredict
from src.network.output import cal_scores_max_min as ref_scores, merge_split_vcfs as ref_merge


class Injected:
    needs_images = False

    def __init__(self, preds):
        self.preds, self.i = preds, 0

    def __call__(self, _images):
        cl, pr = self.preds[self.i]
        self.i += 1
        return None, cl, pr


def one_case(seed):
    rng = np.random.default_rng(seed)
    contigs = [("c%d" % i, int(rng.integers(150_000, 380_000))) for i in range(int(rng.integers(1, 4)))]
    cfg = synth.SimConfig(contigs=contigs, coverage=float(rng.choice([8, 14, 22])), read_len_mean=float(rng.choice([5000, 9000])),
                          read_len_sd=1500.0, err_rate=float(rng.choice([0.0, 0.004])), sv_spacing=float(rng.choice([4000, 9000])),
                          sv_min_gap=int(rng.choice([3000, 8000])), sv_max=int(rng.choice([1500, 6000])), inline_max=int(rng.choice([500, 1500])),
                          het_frac=float(rng.choice([0.0, 0.5, 1.0])), seed=int(seed), sv_mix=MIXES[int(rng.integers(0, len(MIXES)))])
    table, genome, _svs = synth.simulate(cfg)
    over = dict(min_support=int(rng.choice([1, 2, 3, 5])), batch_size=int(rng.choice([1, 7, 64, 128])), qname=bool(rng.random() < 0.3),
                min_gt_depth=int(rng.choice([1, 4, 10])), homo_thresh=float(rng.choice([0.6, 0.8])), hete_thresh=float(rng.choice([0.2, 0.35])),
                max_sv_size=int(rng.choice([2000, 1000000])), sample="S%d" % seed)
    window = int(rng.choice([100_000, 10_000_000]))
    sharp = float(rng.choice([0.3, 2.0, 6.0]))          # how decisive the random classifier is
    out = tempfile.mkdtemp()
    try:
        genome_path = os.path.join(out, "genome.fa")
        bam.write_fasta(genome_path, genome)
        refdriver.FASTAS.clear(); refdriver.DATASETS.clear()
        refdriver.FASTAS[genome_path] = genome
        refdriver.DATASETS["sample.bam"] = table
        seg_dir, rdir, pdir = (os.path.join(out, d) for d in ("segments", "ref_pred", "own_pred"))
        for d in (seg_dir, rdir, pdir):
            os.mkdir(d)
        ropts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", window_size=window,
                                          model_path="unused.ckpt", **over)
        popts = helpers.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", window_size=window,
                                        source_version="1.4", **over)
        fasta = bam.Fasta(sequences=genome)
        scan = helpers.oracle_scan(table, 50)
        chroms = []
        for chrom, clen in contigs:
            text, part, pos = "", 0, 0
            while pos < clen:
                end = min(clen, pos + window)
                sample = Sample.with_scan(table, fasta, 50, scan)
                _s, clusters = detect_window(popts, sample, chrom, pos, end, part)
                text += "".join(p.text() for p in collect_pair_lines(clusters, popts))
                part += 1
                pos = end
            if text:
                chroms.append(chrom)
                with open(os.path.join(seg_dir, chrom + ".segments.all.bed"), "w") as f:
                    f.write(text)
        if not chroms:
            return None, 0
        n_rec = 0
        sample = Sample.with_scan(table, fasta, 50, scan)
        for chrom in chroms:
            bed = os.path.join(seg_dir, chrom + ".segments.all.bed")
            preds = []
            crng = np.random.default_rng(seed * 1000 + len(preds) + hash(chrom) % 97)

            def fn(batch, preds=preds, crng=crng):
                n = np.asarray(batch).shape[0]
                logits = crng.standard_normal((n, 5)).astype(np.float32) * np.float32(sharp)
                z = logits - logits.max(1, keepdims=True)
                e = np.exp(z)
                prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
                cl = logits.argmax(1)
                preds.append((cl.copy(), prob.copy()))
                return logits, cl, prob
            refdriver.PREDICTOR["fn"] = fn
            rprefix = os.path.join(rdir, "%s.predict.s%d" % (chrom, over["min_support"]))
            RefPredict(chrom, bed).run(rprefix, ropts)
            pprefix = os.path.join(pdir, "%s.predict.s%d" % (chrom, over["min_support"]))
            Predict(chrom, bed).run(pprefix, popts, classifier=Injected(preds), sample=sample)
            for ext in (".vcf", ".score.txt"):
                a, b = open(rprefix + ext).read(), open(pprefix + ext).read()
                if a != b:
                    return "MISMATCH seed %d %s%s opts %s: ref %d lines, own %d lines" % (seed, chrom, ext, over, a.count("\n"), b.count("\n")), n_rec
            n_rec += open(rprefix + ".vcf").read().count("\n")
        rs = ref_scores(rdir)
        ps = output.cal_scores_max_min(pdir)
        if list(rs) != list(ps):
            return "MISMATCH seed %d score lists" % seed, n_rec
        if len(rs) == 0:
            return None, n_rec
        mx, mn = np.max(rs), np.min(rs)
        ref_merge(rdir, os.path.join(out, "ref.vcf"), mx, mn, chroms, ropts)
        output.merge_split_vcfs(pdir, os.path.join(out, "own.vcf"), np.max(ps), np.min(ps), chroms, popts, fasta=fasta)
        a, b = open(os.path.join(out, "ref.vcf")).read(), open(os.path.join(out, "own.vcf")).read()
        if a != b:
            la, lb = a.splitlines(), b.splitlines()
            first = next((i for i, (x, y) in enumerate(zip(la, lb)) if x != y), min(len(la), len(lb)))
            return "MISMATCH seed %d merged VCF opts %s at line %d:\n  ref %s\n  own %s" % (
                seed, over, first, la[first] if first < len(la) else None, lb[first] if first < len(lb) else None), n_rec
        return None, n_rec
    finally:
        shutil.rmtree(out)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = total = 0
    for seed in range(first, first + n):
        try:
            msg, recs = one_case(seed)
        except Exception:
            msg, recs = "EXCEPTION seed %d\n%s" % (seed, traceback.format_exc()), 0
        total += recs
        if msg:
            bad += 1
            print(msg, flush=True)
    print("%d cases, %d mismatching, %d VCF records compared" % (n, bad, total))
