#!/usr/bin/env python3
"""Generates tests/golden/graph_small.* : the REFERENCE run with --graph (src/collection/graph.py) on a small sample with
nested / complex SVs whose primary records carry the read bases:
  * run_detect per window with options.graph -> graphs/{contig}-{cstart}-{cend}/{read}.gfa (generate_graph :303-491 called
    from collect_signatures.py:234-306, written by output_clusters.py:57-67);
  * Predict.run (TensorFlow session replaced by the pseudo-classifier of make_predict_fixture) + merge_split_vcfs;
  * collect_csv_same_format (:518-676) -> graph VCF, per-record .gfa, exact / symmetric match summaries.
pysam.VariantFile is a stand-in (refdriver.VariantFile: text round trip + htslib's PASS header line).  This container only."""
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
from src.collection.graph import collect_csv_same_format as ref_graph_vcf  # noqa: E402
from src.collection import graph as ref_graph  # noqa: E402
from src.network.predict import Predict as RefPredict  # noqa: E402
from src.network.output import cal_scores_max_min as ref_scores, merge_split_vcfs as ref_merge  # noqa: E402

WINDOW = 90_000


def tree(root):
    out = {}
    for d, _dirs, files in os.walk(root):
        for name in files:
            p = os.path.join(d, name)
            out[os.path.relpath(p, root)] = open(p).read()
    return out


def iso_cases(tmp):
    """Random small graphs and their mirror images through the reference's parse_gfa_file / graph_is_same_as /
    parse_graph_features / classify_graphs (the symmetric branch, :236-262, does not occur in the small sample)."""
    rng = np.random.default_rng(5)
    texts = []
    for _ in range(14):
        ns, ni = int(rng.integers(2, 5)), int(rng.integers(0, 3))
        order = ["S%d" % i for i in range(ns)] + ["I%d" % i for i in range(ni)]
        lines = []
        for nid in order:
            line = "S\t%s\tACGT\tSN:Z:chrT\tSO:i:%d\tSR:i:0\tLN:i:4" % (nid, int(rng.integers(100, 999)))
            if nid[0] == "I" and rng.random() < 0.5:
                line += "\tDP:S:S0:%d" % int(rng.integers(100, 999))
            lines.append(line)
        walk = list(rng.permutation(order))
        rev = {nid: bool(rng.random() < 0.3) for nid in order}
        links = [(a, rev[a], b, rev[b]) for a, b in zip(walk, walk[1:])]
        body = ["L\t%s\t%s\t%s\t%s\t0M\tSR:i:0" % (a, "-" if ra else "+", b, "-" if rb else "+") for a, ra, b, rb in links]
        texts.append("\n".join(lines + body) + "\n")
        # the same graph read from the other end: ids mirrored per kind, links reversed
        count = {"S": ns, "I": ni}
        mir = {nid: "%s%d" % (nid[0], count[nid[0]] - int(nid[1:]) - 1) for nid in order}
        mlines = [l.replace("\t%s\t" % l.split("\t")[1], "\t%s\t" % mir[l.split("\t")[1]], 1) for l in lines]
        mbody = ["L\t%s\t%s\t%s\t%s\t0M\tSR:i:0" % (mir[b], "-" if rb else "+", mir[a], "-" if ra else "+") for a, ra, b, rb in reversed(links)]
        texts.append("\n".join(mlines + mbody) + "\n")
    texts += texts[:3]                                          # exact repeats for classify_graphs
    graphs = []
    for i, t in enumerate(texts):
        p = os.path.join(tmp, "iso%d.gfa" % i)
        with open(p, "w") as f:
            f.write(t)
        graphs.append(p)
    load = ref_graph.parse_gfa_file
    n = len(texts)
    res = {"gfas": texts, "plain": [], "strict": [], "symmetry": [], "features": [list(ref_graph.parse_graph_features(load(p))) for p in graphs]}
    for i in range(n):
        res["plain"].append([bool(ref_graph.graph_is_same_as(load(graphs[i]), load(graphs[j]))) for j in range(n)])
        res["strict"].append([bool(ref_graph.graph_is_same_as(load(graphs[i]), load(graphs[j]), strict=True)) for j in range(n)])
        res["symmetry"].append([bool(ref_graph.graph_is_same_as(load(graphs[i]), load(graphs[j]), strict=False, symmetry=True)) for j in range(n)])
    ranked = ref_graph.classify_graphs([load(p) for p in graphs])
    res["classified"] = [[ref_graph.parse_graph_features(g)[2], g.appear_time] for g in ranked]
    res["rewritten"] = []
    for i, p in enumerate(graphs[:6]):                          # parse -> write round trip (the per-record files of step 3)
        q = os.path.join(tmp, "rw%d.gfa" % i)
        pos, ids, links = ref_graph.write_graph_to_file(load(p), q)
        res["rewritten"].append({"text": open(q).read(), "positions": sorted(str(v) for v in pos), "ids": ids, "links": links})
    print("iso cases", n, "symmetric pairs", sum(sum(r) for r in res["symmetry"]), "strict pairs", sum(sum(r) for r in res["strict"]))
    return res


def main():
    cfg = synth.SimConfig(contigs=[("chrG", 180_000)], coverage=9, read_len_mean=4200, read_len_sd=600, err_rate=0.003,
                          sv_spacing=2_500, sv_min_gap=4_000, sv_min=80, sv_max=700, inline_max=300, seed=41,
                          sv_mix=(("INV", 0.2), ("DUP", 0.15), ("dDUP", 0.2), ("DELINV", 0.2), ("DEL", 0.1), ("INS", 0.1), ("cINS", 0.05)))
    table, genome, svs = synth.simulate(cfg, with_seq=True)
    bam_path = os.path.join(HERE, "graph_small.bam")
    bam.write_bam(bam_path, table, level=9)
    with open(os.path.join(HERE, "graph_small.fa.gz"), "wb") as _raw, gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as f:   # mtime 0: regenerates byte for byte
        for name, seq in genome.items():
            f.write(b">" + name.encode() + b"\n" + seq + b"\n")
    out = tempfile.mkdtemp()
    seg_dir, pred_dir, graph_dir = (os.path.join(out, d) for d in ("segments", "predict_results", "graphs"))
    for d in (seg_dir, pred_dir, graph_dir):
        os.mkdir(d)
    genome_path = os.path.join(out, "genome.fa")
    bam.write_fasta(genome_path, genome)
    refdriver.DATASETS["sample.bam"] = bam.read_bam(bam_path, with_seq=True)
    refdriver.FASTAS[genome_path] = genome
    opts = refdriver.default_options(out_path=out, genome=genome_path, bam_path="sample.bam", min_support=3, window_size=WINDOW,
                                     batch_size=64, model_path="unused.ckpt", sample="HGg", graph=True, qname=True)
    expected = {"window": WINDOW, "min_support": 3, "batch_size": 64, "sample": "HGg", "qname": True, "windows": []}
    chrom, clen = cfg.contigs[0]
    part, pos = 0, 0
    while pos < clen:
        end = min(clen, pos + WINDOW)
        err = ref_run.run_detect(opts, "sample.bam", chrom, part, pos, end)
        assert err is None, err
        p = os.path.join(seg_dir, "%s.segments.%d.bed" % (chrom, part))
        expected["windows"].append({"start": pos, "end": end, "tsv": open(p).read() if os.path.exists(p) else ""})
        part, pos = part + 1, end
    expected["read_graphs"] = tree(graph_dir)
    bed = os.path.join(seg_dir, chrom + ".segments.all.bed")
    with open(bed, "w") as f:
        f.write("".join(w["tsv"] for w in expected["windows"]))
    preds = []

    def fn(batch):
        lo, cl, pr = pseudo_classifier(batch)
        preds.append((cl.copy(), pr.copy()))
        return lo, cl, pr
    refdriver.PREDICTOR["fn"] = fn
    prefix = os.path.join(pred_dir, "%s.predict.s%d" % (chrom, opts.min_support))
    RefPredict(chrom, bed).run(prefix, opts)
    scores = ref_scores(pred_dir)
    merged = os.path.join(out, "HGg.svision.s3.vcf")
    ref_merge(pred_dir, merged, np.max(scores), np.min(scores), [chrom], opts)
    expected["merged_vcf"] = open(merged).read()
    expected["classes"] = np.concatenate([p[0] for p in preds]).tolist()
    expected["probs"] = np.concatenate([p[1] for p in preds]).astype(np.float32).view(np.uint32).tolist()
    exact, symmetric = ref_graph_vcf(graph_dir, merged, opts)
    expected["graph_vcf"] = open(os.path.join(out, "HGg.svision.s3.graph.vcf")).read()
    expected["exactly_match"] = open(os.path.join(out, "HGg.graph_exactly_match.txt")).read()
    expected["symmetry_match"] = open(os.path.join(out, "HGg.graph_symmetry_match.txt")).read()
    expected["record_graphs"] = {k: v for k, v in tree(graph_dir).items() if os.sep not in k}
    expected["iso"] = iso_cases(out)
    shutil.rmtree(out)
    n_csv = sum(1 for l in expected["merged_vcf"].splitlines() if not l.startswith("#") and "CSV" in l)
    print("records", len(table), "per-read graphs", len(expected["read_graphs"]), "vcf records",
          sum(1 for l in expected["merged_vcf"].splitlines() if not l.startswith("#")), "complex", n_csv,
          "distinct graphs", len(exact), "symmetric", len(symmetric),
          "graphs with inserted nodes", sum(1 for t in expected["read_graphs"].values() if "\tI0\t" in t),
          "with dup nodes", sum(1 for t in expected["read_graphs"].values() if "\tDP:S:" in t))
    assert n_csv >= 3 and len(exact) >= 2
    import io
    with open(os.path.join(HERE, "graph_small.expected.json.gz"), "wb") as _raw, \
            gzip.GzipFile(filename="", mode="wb", fileobj=_raw, mtime=0, compresslevel=9) as _gz, io.TextIOWrapper(_gz) as f:
        json.dump(expected, f, sort_keys=True)


if __name__ == "__main__":
    main()
