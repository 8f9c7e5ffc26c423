"""--graph against a reference-run fixture (tests/golden/make_graph_fixture.py): the per-read breakpoint
graphs written during collection, the graph-annotated VCF with the per-record graphs and the two match summaries of
step 3, and the graph comparison primitives on random graphs and their mirror images."""
import gzip
import json
import os

import pytest

from svision_amd.collection import graph
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.io import bam
from svision_amd.sample import Sample
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with gzip.open(os.path.join(helpers.GOLDEN, "graph_small.expected.json.gz"), "rt") as f:
        return json.load(f)


def _tree(root):
    out = {}
    for d, _dirs, files in os.walk(root):
        for name in files:
            p = os.path.join(d, name)
            with open(p) as f:
                out[os.path.relpath(p, root)] = f.read()
    return out


def _options(expected, out, **over):
    kw = dict(out_path=str(out), min_support=expected["min_support"], batch_size=expected["batch_size"],
              window_size=expected["window"], sample=expected["sample"], graph=True, qname=True)
    kw.update(over)
    return helpers.default_options(**kw)


def test_read_graphs_of_the_collection_match_reference(expected, tmp_path, oracle_lib):
    table = bam.read_bam(os.path.join(helpers.GOLDEN, "graph_small.bam"), with_seq=True)
    sample = Sample.with_scan(table, helpers.load_golden_fasta("graph_small.fa.gz"), 50, helpers.oracle_scan(table, 50))
    os.mkdir(tmp_path / "graphs")
    opts = _options(expected, tmp_path)
    for part, w in enumerate(expected["windows"]):
        _sigs, clusters = detect_window(opts, sample, "chrG", w["start"], w["end"], part)
        lines = collect_pair_lines(clusters, opts)
        assert "".join(ln.text() for ln in lines) == w["tsv"]
    got = _tree(tmp_path / "graphs")
    assert sorted(got) == sorted(expected["read_graphs"])
    for name, text in expected["read_graphs"].items():
        assert got[name] == text, name
    assert sum(1 for t in got.values() if "\tDP:S:" in t) > 10 and sum(1 for t in got.values() if "\tI0\t" in t) > 10


def test_graph_off_leaves_no_graphs_and_the_same_tsv(expected, tmp_path, oracle_lib):
    table = bam.read_bam(os.path.join(helpers.GOLDEN, "graph_small.bam"))
    sample = Sample.with_scan(table, helpers.load_golden_fasta("graph_small.fa.gz"), 50, helpers.oracle_scan(table, 50))
    opts = _options(expected, tmp_path, graph=False)
    w = expected["windows"][0]
    _sigs, clusters = detect_window(opts, sample, "chrG", w["start"], w["end"], 0)
    assert "".join(ln.text() for ln in collect_pair_lines(clusters, opts)) == w["tsv"]
    assert not os.path.exists(tmp_path / "graphs")


def test_graph_vcf_and_summaries_match_reference(expected, tmp_path):
    gdir = tmp_path / "graphs"
    os.mkdir(gdir)
    for name, text in expected["read_graphs"].items():
        os.makedirs(gdir / os.path.dirname(name), exist_ok=True)
        with open(gdir / name, "w") as f:
            f.write(text)
    vcf = tmp_path / "in.vcf"
    with open(vcf, "w") as f:
        f.write(expected["merged_vcf"])
    opts = _options(expected, tmp_path)
    exact, symmetric = graph.annotate_vcf_with_graphs(str(gdir), str(vcf), opts)
    name = "%s.svision.s%d.graph.vcf" % (expected["sample"], expected["min_support"])
    assert open(tmp_path / name).read() == expected["graph_vcf"]
    assert open(tmp_path / ("%s.graph_exactly_match.txt" % expected["sample"])).read() == expected["exactly_match"]
    assert open(tmp_path / ("%s.graph_symmetry_match.txt" % expected["sample"])).read() == expected["symmetry_match"]
    got = {k: v for k, v in _tree(gdir).items() if os.sep not in k}
    assert got == expected["record_graphs"]
    assert len(exact) >= 2 and sum("GraphID=-1" not in l for l in expected["graph_vcf"].splitlines() if not l.startswith("#")) >= 3


def test_complex_record_without_reads_field_fails_like_upstream(expected, tmp_path):
    """--graph without --qname: the merged VCF has no READS field and step 3 of the reference dies on the first complex
    record (record.info['READS'], graph.py:579)."""
    gdir = tmp_path / "graphs"
    os.mkdir(gdir)
    for name in expected["read_graphs"]:
        os.makedirs(gdir / os.path.dirname(name), exist_ok=True)
    vcf = tmp_path / "in.vcf"
    with open(vcf, "w") as f:
        for line in expected["merged_vcf"].splitlines(True):
            f.write(line if line.startswith("#") else ";".join(kv for kv in line.split(";") if not kv.startswith("READS=")))
    with pytest.raises(KeyError):
        graph.annotate_vcf_with_graphs(str(gdir), str(vcf), _options(expected, tmp_path))


def test_graph_comparison_primitives_match_reference(expected, tmp_path):
    iso = expected["iso"]
    paths = []
    for i, text in enumerate(iso["gfas"]):
        p = tmp_path / ("g%d.gfa" % i)
        with open(p, "w") as f:
            f.write(text)
        paths.append(str(p))
    n = len(paths)
    for i in range(n):
        for j in range(n):
            a, b = graph.read_gfa(paths[i]), graph.read_gfa(paths[j])
            assert graph.same_graph(a, b) == iso["plain"][i][j], (i, j)
            assert graph.same_graph(a, b, strict=True) == iso["strict"][i][j], (i, j)
            assert graph.same_graph(a, b, strict=False, symmetry=True) == iso["symmetry"][i][j], (i, j)
    assert sum(sum(r) for r in iso["symmetry"]) > n                      # mirror pairs are found, not just i == j
    assert [list(graph.graph_features(graph.read_gfa(p))) for p in paths] == iso["features"]
    ranked = graph.most_common_graphs([graph.read_gfa(p) for p in paths])
    assert [[graph.graph_features(g)[2], g.appear_time] for g in ranked] == iso["classified"]
    for i, want in enumerate(iso["rewritten"]):
        q = tmp_path / ("rw%d.gfa" % i)
        pos, ids, links = graph.write_gfa(graph.read_gfa(paths[i]), str(q))
        assert open(q).read() == want["text"] and sorted(str(v) for v in pos) == want["positions"]
        assert ids == want["ids"] and links == want["links"]


def test_cli_with_graph_reproduces_reference_outputs(expected, tmp_path, oracle_lib):
    """The whole driver with --graph --qname (CNN outputs injected, as in tests/test_cli_e2e.py): collection writes the
    per-read graphs, step 3 turns them into the reference's graph VCF / per-record graphs / summaries, the per-site
    folders and the plain VCF are removed (SVision:341-359)."""
    import numpy as np
    from svision_amd import cli
    from tests.test_cli_e2e import ChromInjected
    table = bam.read_bam(os.path.join(helpers.GOLDEN, "graph_small.bam"), with_seq=True)
    sample = Sample.with_scan(table, helpers.load_golden_fasta("graph_small.fa.gz"), 50, helpers.oracle_scan(table, 50))
    out = str(tmp_path / "out")
    opts = cli.parse_arguments(["-o", out, "-b", "/virtual/graph.bam", "-m", "/virtual/model.ckpt", "-g", "/virtual/genome.fa",
                                "-n", expected["sample"], "-s", str(expected["min_support"]), "--window_size", str(expected["window"]),
                                "--batch_size", str(expected["batch_size"]), "--graph", "--qname"])
    case = {"batch_size": expected["batch_size"], "chroms": {"chrG": {"classes": expected["classes"], "probs": expected["probs"]}}}
    merged = cli.run(opts, sample=sample, classifier=ChromInjected(case, ["chrG"]))
    assert os.path.basename(merged) == "HGg.svision.s3.graph.vcf" and open(merged).read() == expected["graph_vcf"]
    assert not os.path.exists(os.path.join(out, "HGg.svision.s3.vcf"))
    assert open(os.path.join(out, "HGg.graph_exactly_match.txt")).read() == expected["exactly_match"]
    assert open(os.path.join(out, "HGg.graph_symmetry_match.txt")).read() == expected["symmetry_match"]
    assert _tree(os.path.join(out, "graphs")) == expected["record_graphs"]       # no per-site folder left
