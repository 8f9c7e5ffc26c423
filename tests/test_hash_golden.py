"""--hash re-aligner vs the reference's own hashplot_unmapped, imported unmodified when
the fixture was generated (tests/golden/make_hash_fixture.py)."""
import json
import os

from svision_amd.segmentplot.run_hash_lineplot import hashplot_unmapped
from tests import helpers


def test_hashplot_unmapped_matches_reference():
    with open(os.path.join(helpers.GOLDEN, "hash_small.expected.json")) as f:
        cases = json.load(f)
    hits = 0
    for c in cases:
        main, segs = hashplot_unmapped(c["ref"], c["seq"], 10, 50)
        assert main is None
        got = [[s.xStart(), s.xEnd(), s.yStart(), s.yEnd(), bool(s.forward())] for s in segs]
        assert got == c["segs"]
        hits += bool(got)
    assert hits > 80


def test_collection_with_hash_matches_reference(oracle_lib):
    """run_detect with --hash (and without) on a sample with real read bases: signatures incl. the helper
    segments found by the re-aligner, and the TSV, equal the reference's (tests/golden/make_hash_collect_fixture.py)."""
    import gzip
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    with open(os.path.join(helpers.GOLDEN, "hash_collect.expected.json")) as f:
        expected = json.load(f)
    fasta = helpers.load_golden_fasta("hash_collect.fa.gz")
    for w in expected["windows"]:
        table = bam.read_bam(os.path.join(helpers.GOLDEN, "hash_collect.bam"), with_seq=True)
        sample = Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50))
        opts = helpers.default_options(min_support=3, hash=w["hash"])
        sigs, clusters = detect_window(opts, sample, "chrH", 0, 160_000)
        got = [[s.type, s.tstart, s.tend, s.qname, s.bkps, s.mechanism,
                [[a.q_start, a.q_end, a.ref_start, a.ref_end, bool(a.is_reverse)] for a in s.sorted_aligns]] for s in sigs]
        assert got == w["signatures"]
        assert "".join(p.text() for p in collect_pair_lines(clusters, opts)) == w["tsv"]
    assert sum(1 for d in expected["windows"][0]["signatures"] if len(d[6]) > 2) == 53
