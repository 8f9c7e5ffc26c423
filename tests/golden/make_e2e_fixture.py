#!/usr/bin/env python3
"""Generates tests/golden/e2e_small.expected.json.gz: the REFERENCE'S OWN DRIVER SCRIPT (/root/reference/SVision, run
as ``__main__`` through runpy with its real argument parser) from BAM + FASTA to the merged VCF, with real CNN
arithmetic: the TensorFlow session of Predict.run (/root/reference/src/network/predict.py:206-210) is the NumPy fp32
restatement of the reference's AlexNet graph (oracle/alexnet_ref.py) on fixed weights -- the weights tests/e2e_weights.py
rebuilds from a seed plus the fc8 calibration stored in the fixture, and writes as a TF checkpoint for the product's
``-m``.  So every step the reference itself performs between the BAM and the VCF -- tasking (SVision:164-234), run_detect
per window (:259-281), ``cat`` (:284-288), BatchGenerator + PlotSingleImg, sess.run -> round(softmax, 2) -> vote
(predict.py:206-300), write_results_to_vcf + genotyper (output.py:469-598), score range + merge (SVision:331-339,
output.py:251-348) -- runs unmodified, in the reference's own order, on:

  collect   collect_small.bam   -s 3 --window_size 150000 --batch_size 128
  boundary  boundary_small.bam  -s 4 --window_size 100000 --batch_size 64 --qname     (sites spanning window boundaries)
  ont       ont_small.bam       -s 2 --window_size 100000 --batch_size 128             (ONT-like reads)
  contig    collect_small.bam   --contig --batch_size 64                               (min_support 1, one task per contig)

Third-party stand-ins are refdriver.py's (pysam field semantics, cv2.line = oracle clipLine + LineIterator); the only
other substitution is multiprocessing.Pool -> an in-process pool (the stand-ins live in this process).
Stored per case: the command line, per chromosome the concatenated TSV, VCF body, score file and every image's
(argmax, softmax bits) as the session returned them, and the merged VCF.  Run in this container only (~5 min)."""
import gzip
import json
import logging
import multiprocessing
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch  # noqa: F401  (before the tensorflow stand-in is installed: torch's import inspects every module's __file__)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdriver  # noqa: E402

refdriver.install_stubs()
from oracle import alexnet_ref, encode_ref  # noqa: E402
from svision_amd.io import bam  # noqa: E402
from svision_amd.network.create_batch import parse_data_fields  # noqa: E402
from tests import e2e_weights, helpers  # noqa: E402

CASES = (
    ("collect", "collect_small", ["-s", "3", "--window_size", "150000", "--batch_size", "128"]),
    ("boundary", "boundary_small", ["-s", "4", "--window_size", "100000", "--batch_size", "64", "--qname"]),
    ("ont", "ont_small", ["-s", "2", "--window_size", "100000", "--batch_size", "128"]),
    ("contig", "collect_small", ["--contig", "--batch_size", "64"]),
)
SAMPLE_NAME = "HGe2e"


class InlinePool:
    """multiprocessing.Pool(processes=N) of SVision:261,311 run in this process, task by task, in submission order."""

    def __init__(self, processes=None):
        pass

    def apply_async(self, fn, args=()):
        value = fn(*args)
        return types.SimpleNamespace(get=lambda: value)

    def close(self):
        pass

    def join(self):
        pass


def calibrate_fc8(params):
    """fc8 of the seeded weights rescaled per class so that the 5 logits vary over the candidate images (random weights
    answer every similarity image with the same class: the vote, SVTYPE and QUAL paths would see one value).  Computed
    from the first 256 lines of the `collect` TSV; the per-class scale and bias are stored in the fixture."""
    with open(os.path.join(HERE, "collect_small.expected.json")) as f:
        exp = json.load(f)
    lines = [l for w in exp["windows"] if w["min_support"] == 3 for l in w["tsv"].splitlines()][:256]
    rec = np.asarray([parse_data_fields(l.split("\t")[1:13]) for l in lines], np.int32)
    logits = alexnet_ref.forward(params, encode_ref.encode_records(rec)).astype(np.float64)
    scale = (1.5 / logits.std(0)).astype(np.float32)
    bias = (params["fc8/biases"].astype(np.float64) * scale - logits.mean(0) * scale).astype(np.float32)
    return scale, bias


def run_reference_driver(argv):
    saved_argv, saved_pool = sys.argv, multiprocessing.Pool
    root = logging.getLogger()
    before = list(root.handlers)
    sys.argv = ["SVision"] + argv                     # parse_arguments' default is bound when the script is executed
    multiprocessing.Pool = InlinePool
    try:
        runpy.run_path(os.path.join(refdriver.REF_ROOT, "SVision"), run_name="__main__")
    finally:
        sys.argv, multiprocessing.Pool = saved_argv, saved_pool
        for h in list(root.handlers):
            if h not in before:
                root.removeHandler(h)
                h.close()


def main():
    params = alexnet_ref.random_params(seed=e2e_weights.SEED)
    scale, bias = calibrate_fc8(params)
    params = e2e_weights.apply_calibration(params, scale, bias)
    expected = {"seed": e2e_weights.SEED, "fc8_scale": scale.view(np.uint32).tolist(), "fc8_bias": bias.view(np.uint32).tolist(),
                "weights_crc": e2e_weights.params_crc(params), "sample": SAMPLE_NAME, "cases": {}}
    for name, data, extra in CASES:
        out = tempfile.mkdtemp()
        bam_path = os.path.join(HERE, data + ".bam")
        fasta = helpers.load_golden_fasta(data + ".fa.gz")
        genome = {n: fasta._seq[n] for n in fasta.references}
        genome_path = os.path.join(out, "genome.fa")
        bam.write_fasta(genome_path, genome)
        refdriver.DATASETS[bam_path] = bam.read_bam(bam_path)
        refdriver.FASTAS[genome_path] = genome
        preds = []

        def session_run(batch, preds=preds):
            lo, cl, pr = alexnet_ref.predict(params, np.asarray(batch, np.float32))
            preds.append((cl.copy(), pr.copy()))
            return lo, cl, pr
        refdriver.PREDICTOR["fn"] = session_run
        work = os.path.join(out, "work")
        argv = ["-o", work, "-b", bam_path, "-m", os.path.join(out, "model.ckpt"), "-g", genome_path, "-n", SAMPLE_NAME, "--debug"] + extra
        run_reference_driver(argv)
        min_support = 1 if "--contig" in extra else int(extra[extra.index("-s") + 1])
        batch = int(extra[extra.index("--batch_size") + 1])
        merged = open(os.path.join(work, "%s.svision.s%d.vcf" % (SAMPLE_NAME, min_support))).read()
        case = {"data": data, "args": extra, "min_support": min_support, "batch_size": batch, "merged_vcf": merged, "chroms": {}, "chrom_order": []}
        k = 0
        for chrom in refdriver.DATASETS[bam_path].references:       # task-dict order = BAM header order (SVision:172-201)
            bed = os.path.join(work, "segments", chrom + ".segments.all.bed")
            if not os.path.exists(bed):
                continue
            tsv = open(bed).read()
            n_batches = -(-tsv.count("\n") // batch)
            mine = preds[k:k + n_batches]
            k += n_batches
            prefix = os.path.join(work, "predict_results", "%s.predict.s%d" % (chrom, min_support))
            case["chrom_order"].append(chrom)
            case["chroms"][chrom] = {
                "tsv": tsv, "vcf": open(prefix + ".vcf").read(), "score": open(prefix + ".score.txt").read(),
                "classes": np.concatenate([p[0] for p in mine]).tolist() if mine else [],
                "probs": np.concatenate([p[1] for p in mine]).astype(np.float32).view(np.uint32).ravel().tolist() if mine else []}
        assert k == len(preds), (k, len(preds))
        expected["cases"][name] = case
        body = [l for l in merged.splitlines() if not l.startswith("#")]
        cls = np.concatenate([np.asarray(c["classes"]) for c in case["chroms"].values()])
        print(name, "images", sum(c["tsv"].count("\n") for c in case["chroms"].values()), "records", len(body),
              "classes", np.bincount(cls, minlength=5).tolist(), "QUAL", sorted({l.split("\t")[5] for l in body})[:12])
        shutil.rmtree(out)
    raw = json.dumps(expected).encode()
    with open(os.path.join(HERE, "e2e_small.expected.json.gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", compresslevel=9, mtime=0, filename="") as g:     # mtime 0: regenerates byte for byte
            g.write(raw)


if __name__ == "__main__":
    main()
