"""A site that spans a collection-window boundary is ONE site (tests/golden/make_boundary_fixture.py).

The reference votes over the concatenation of the windows' TSVs (SVision:284-288, predict.py:235-247), so a region
string that closes window k and opens window k+1 -- reads overlapping the boundary are collected by both windows --
yields a single VCF record.  The fixture (4 such sites) holds the reference's per-window TSVs, its per-chromosome VCF
bodies / scores and its merged VCF; the CNN outputs are injected.  Checked here: the per-chromosome voter
(Predict.run), and the per-window votes of the pooled / streaming pipelines stitched by ChromosomeVote."""
import io
import json
import os

import numpy as np
import pytest

from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.network import output
from svision_amd.pipeline import WindowResult, _vote, stitch_windows
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(helpers.GOLDEN, "boundary_small.expected.json")) as f:
        return json.load(f)


def _options(expected, **over):
    return helpers.default_options(min_support=expected["min_support"], batch_size=expected["batch_size"], window_size=expected["window"],
                                   sample="HGb", source_version="1.4", **over)


def test_fixture_has_sites_on_window_boundaries(expected):
    shared = 0
    for c in expected["chroms"].values():
        tsvs = [w["tsv"] for w in c["windows"]]
        for a, b in zip(tsvs, tsvs[1:]):
            if a and b and a.splitlines()[-1].split("\t")[0] == b.splitlines()[0].split("\t")[0]:
                shared += 1
    assert shared == expected["boundary_sites"] >= 3


def _window_results(expected, sample_factory, chrom):
    """Per-window collection (product host code) + per-window vote with the fixture's CNN outputs, as the helpers do."""
    c = expected["chroms"][chrom]
    classes = np.asarray(c["classes"], np.int64)
    probs = np.asarray(c["probs"], np.uint32).view(np.float32).reshape(-1, 5)
    opts = _options(expected)
    results, lo = [], 0
    for part, w in enumerate(c["windows"]):
        sample = sample_factory()
        _sigs, clusters = detect_window(opts, sample, chrom, w["start"], w["end"], part)
        lines = collect_pair_lines(clusters, opts)
        assert "".join(ln.text() for ln in lines) == w["tsv"]                  # the encode side equals the reference's
        res = WindowResult()
        res.chrom, res.start, res.end = chrom, w["start"], w["end"]
        res.vcf, res.scores, res.n_sites, res.head, res.tail = _vote(sample, opts, chrom, lines, classes[lo:lo + len(lines)], probs[lo:lo + len(lines)],
                                                                     w["start"], w["end"])
        lo += len(lines)
        results.append(res)
    return results, opts


def test_stitched_window_votes_equal_the_reference_chromosome_vote(expected, oracle_lib, tmp_path):
    factory = lambda: helpers.golden_sample(50, name="boundary_small")       # noqa: E731
    pred_dir = tmp_path / "pred"
    pred_dir.mkdir()
    held = 0
    for chrom in expected["chrom_order"]:
        results, opts = _window_results(expected, factory, chrom)
        held += sum(1 for r in results if r.head) + sum(1 for r in results if r.tail)
        texts = stitch_windows(results, opts, factory())
        vcf, score = texts.get(chrom, ("", ""))
        assert vcf == expected["chroms"][chrom]["vcf"]
        assert score == expected["chroms"][chrom]["score"]
        # without the stitch (every window flushing its own edge sites) the boundary sites come out twice
        naive = "".join(_flush_all(r, opts, factory(), chrom) for r in results)
        assert naive.count("\n") > vcf.count("\n")
        (pred_dir / ("%s.predict.s%d.vcf" % (chrom, opts.min_support))).write_text(vcf)
        (pred_dir / ("%s.predict.s%d.score.txt" % (chrom, opts.min_support))).write_text(score)
    assert held >= 8                                           # the boundary sites (and only sites near a boundary) were held back
    scores = output.cal_scores_max_min(str(pred_dir))
    mx, mn = np.max(scores), np.min(scores)
    assert float(mx) == expected["max_score"] and float(mn) == expected["min_score"]
    merged = str(tmp_path / "merged.vcf")
    opts.out_path = str(tmp_path)
    output.merge_split_vcfs(str(pred_dir), merged, mx, mn, expected["chrom_order"], opts, fasta=factory().fasta)
    assert open(merged).read() == expected["merged_vcf"]


def _flush_all(res, opts, sample, chrom):
    """What a per-window vote that ignores the boundary would write for this window."""
    from svision_amd.network.predict import Predict, SiteVoter
    out, sc = io.StringIO(), io.StringIO()
    v = SiteVoter(Predict(chrom, None), out, sc, opts, sample)
    if res.head:
        v.feed_items(res.head)
    v.close_site()
    out.write(res.vcf)
    if res.tail:
        v.feed_items(res.tail)
    v.close_site()
    return out.getvalue()


def test_predict_run_over_concatenated_bed(expected, oracle_lib, tmp_path):
    """The file-based path (Predict.run over {chrom}.segments.all.bed) on the same fixture."""
    from svision_amd.network.predict import Predict
    from tests.test_predict_golden import Injected
    sample = helpers.golden_sample(50, name="boundary_small")
    opts = _options(expected)
    for chrom in expected["chrom_order"]:
        c = expected["chroms"][chrom]
        bed = tmp_path / (chrom + ".all.bed")
        bed.write_text("".join(w["tsv"] for w in c["windows"]))
        prefix = str(tmp_path / chrom)
        Predict(chrom, str(bed)).run(prefix, opts, classifier=Injected(c["classes"], c["probs"], opts.batch_size), sample=sample)
        assert open(prefix + ".vcf").read() == c["vcf"] and open(prefix + ".score.txt").read() == c["score"]


@pytest.mark.gpu
def test_streaming_and_pooled_cli_paths_agree_on_boundary_sites(tmp_path):
    """-t 1 (one voter per chromosome) and -t 3 (per-window votes in helper processes, stitched) write the same files
    on the device path, window boundaries included."""
    from svision_amd import cli
    from svision_amd.io import bam
    from svision_amd.network import tf_checkpoint as ck
    from oracle import alexnet_ref
    prefix = str(tmp_path / "svision-cnn-model.ckpt")
    ck.write_checkpoint(prefix, alexnet_ref.random_params(seed=7))
    fasta = helpers.load_golden_fasta("boundary_small.fa.gz")
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    outs = []
    for t in (1, 3):
        out = str(tmp_path / ("out%d" % t))
        opts = cli.parse_arguments(["-o", out, "-b", os.path.join(helpers.GOLDEN, "boundary_small.bam"), "-m", prefix, "-g", fa, "-n", "HGb",
                                    "-s", "4", "--window_size", "100000", "--batch_size", "64", "-t", str(t), "--debug"])
        merged = cli.run(opts)
        files = {"merged": open(merged).read()}
        for name in sorted(os.listdir(os.path.join(out, "predict_results"))):
            files[name] = open(os.path.join(out, "predict_results", name)).read()
        for name in sorted(os.listdir(os.path.join(out, "segments"))):
            files[name] = open(os.path.join(out, "segments", name)).read()
        outs.append(files)
    assert outs[0] == outs[1]
    assert outs[0]["merged"].count("\n") > 10


def test_stitched_votes_equal_one_vote_per_chromosome_on_dense_boundaries(oracle_lib):
    """Property test on a sample cut into windows only ~4 read lengths wide (every window edge is near a site): the
    per-window votes -- edge sites held back only within edge_margin of a boundary -- stitched per chromosome write
    exactly what ONE voter over the concatenated lines writes (the semantics pinned to the reference above)."""
    import zlib
    from svision_amd import synth
    from svision_amd.io import bam
    from svision_amd.network.predict import Predict, SiteVoter
    from svision_amd.sample import Sample
    cfg = synth.SimConfig(contigs=[("chrS", 400_000)], coverage=18, read_len_mean=7000, read_len_sd=1500, err_rate=0.004,
                          sv_spacing=4_000, sv_min_gap=5_000, sv_max=3000, inline_max=1200, seed=41)
    table, genome, _ = synth.simulate(cfg)
    fasta = bam.Fasta(sequences=genome)
    scan = helpers.oracle_scan(table, 50)
    opts = helpers.default_options(min_support=3, batch_size=64, window_size=30_000)
    results, all_lines, all_cls, all_prob = [], [], [], []
    held = flushed_edges = 0
    for part, start in enumerate(range(0, 400_000, 30_000)):
        end = min(400_000, start + 30_000)
        sample = Sample.with_scan(table, fasta, 50, scan)
        _s, clusters = detect_window(opts, sample, "chrS", start, end, part)
        lines = collect_pair_lines(clusters, opts)
        h = np.array([zlib.crc32(ln.text().encode()) for ln in lines], np.int64)
        cls = h % 5
        prob = np.full((len(lines), 5), 0.05, np.float32)
        prob[np.arange(len(lines)), cls] = (0.5 + (h % 50) / 100.0).astype(np.float32)
        res = WindowResult()
        res.chrom, res.start, res.end = "chrS", start, end
        res.vcf, res.scores, res.n_sites, res.head, res.tail = _vote(sample, opts, "chrS", lines, cls, prob, start, end)
        held += bool(res.head) + bool(res.tail)
        flushed_edges += (2 if len({ln.region for ln in lines}) > 1 else 1 if lines else 0) - bool(res.head) - bool(res.tail)
        results.append(res)
        all_lines += lines
        all_cls.append(cls)
        all_prob.append(prob)
    sample = Sample.with_scan(table, fasta, 50, scan)
    vcf, score = io.StringIO(), io.StringIO()
    one = SiteVoter(Predict("chrS", None), vcf, score, opts, sample)
    one.feed_batch([ln.label() for ln in all_lines], np.concatenate(all_cls), np.concatenate(all_prob))
    one.finish()
    got_vcf, got_score = stitch_windows(results, opts, sample)["chrS"]
    assert got_vcf == vcf.getvalue() and got_score == score.getvalue()
    assert got_vcf.count("\n") > 20 and held > 10
    regions = [ln.region for ln in all_lines]
    assert len(set(regions)) < sum(r.n_sites for r in results)   # some site really is collected by two windows
