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


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_seed_and_extend_matches_reference_cases():
    """svx_hash_seeds + the host merge == the reference's hashplot_unmapped on the 220 golden cases (one launch for all),
    and its raw hit lists == the host aligner's, in order."""
    from svision_amd import kernels
    from svision_amd.segmentplot import run_hash_lineplot as rh
    from svision_amd.segmentplot.hash_aligner import HashAligner
    with open(os.path.join(helpers.GOLDEN, "hash_small.expected.json")) as f:
        cases = json.load(f)
    got = rh.hashplot_unmapped_batch([(c["ref"], c["seq"]) for c in cases], 10, 50, "cuda:0")
    hits = 0
    for c, segs in zip(cases, got):
        assert segs is not None
        assert [[s.xStart(), s.xEnd(), s.yStart(), s.yEnd(), bool(s.forward())] for s in segs] == c["segs"]
        hits += bool(segs)
    assert hits > 80
    # raw lists of a few cases against the host aligner (self pass and placement pass), order included
    for c in cases[:40]:
        x, y = kernels.pack_bases(c["seq"]), kernels.pack_bases(c["ref"])
        hits_a, hits_b = kernels.hash_seeds([(x, y)], 10, 50, "cuda:0")[0]
        a = HashAligner(10, 50, 0, 2)
        a.run(c["ref"], c["ref"])
        want_a = [[s.yStart(), s.xStart() if s.forward() else (len(y) - 1) - s.xStart(), s._length, int(bool(s.forward()))] for s in a.getSegments()]
        assert hits_a.tolist() == want_a
        b = HashAligner(10, 50, 0, 2)
        b.compareDiffSegs = []                                   # keep every hit
        b.y_hashvalues = a.getHashValues()
        b._align(c["seq"], c["ref"], a.getAvoidKmer())
        want_b = [[s.yStart(), s.xStart() if s.forward() else (len(x) - 1) - s.xStart(), s._length, int(bool(s.forward()))] for s in b.getSegments()]
        assert hits_b.tolist() == want_b
    # sequences outside ACGTN are refused by the packer (host path)
    assert kernels.pack_bases("ACGTnACGT") is not None and kernels.pack_bases("ACGTBDACGT") is None
    # hostile: N runs, a piece shorter than k, a window shorter than k, identical sequences, palindromes
    rng = __import__("numpy").random.default_rng(3)
    def rnd(n, p_n=0.0):
        return "".join(rng.choice(list("ACGTN"), p=[(1 - p_n) / 4] * 4 + [p_n]) for _ in range(n))
    hostile = [(rnd(400), rnd(5)), (rnd(8), rnd(300)), (rnd(300, 0.05), rnd(200, 0.05)), ("ACGT" * 100, "ACGT" * 40)]
    r = rnd(500)
    hostile += [(r, r[100:350]), (r, "".join({"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}[b] for b in reversed(r[50:400])))]
    for (ref, seq), segs in zip(hostile, rh.hashplot_unmapped_batch(hostile, 10, 50, "cuda:0")):
        want = rh._hashplot_host(ref, seq, 10, 50)
        fmt = lambda ss: [[s.xStart(), s.xEnd(), s.yStart(), s.yEnd(), bool(s.forward())] for s in ss]     # noqa: E731
        assert fmt(segs) == fmt(want)


@pytest.mark.gpu
def test_collection_with_hash_on_the_device_matches_reference():
    """run_detect --hash with the device scan AND the device seed-and-extend == the reference's signatures and TSV."""
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    from svision_amd.segmentplot import run_hash_lineplot as rh
    from svision_amd import kernels
    with open(os.path.join(helpers.GOLDEN, "hash_collect.expected.json")) as f:
        expected = json.load(f)
    fasta = helpers.load_golden_fasta("hash_collect.fa.gz")
    calls = []
    orig = kernels.hash_seeds
    kernels.hash_seeds = lambda *a, **kw: (calls.append(len(a[0])), orig(*a, **kw))[1]
    try:
        for w in expected["windows"]:
            table = bam.read_bam(os.path.join(helpers.GOLDEN, "hash_collect.bam"), with_seq=True)
            sample = Sample.from_table(table, fasta, 50, device="cuda:0")
            assert rh.DEVICE is not None
            opts = helpers.default_options(min_support=3, hash=w["hash"])
            sigs, clusters = detect_window(opts, sample, "chrH", 0, 160_000)
            got = [[s.type, s.tstart, s.tend, s.qname, s.bkps, s.mechanism,
                    [[a.q_start, a.q_end, a.ref_start, a.ref_end, bool(a.is_reverse)] for a in s.sorted_aligns]] for s in sigs]
            assert got == w["signatures"]
            assert "".join(p.text() for p in collect_pair_lines(clusters, opts)) == w["tsv"]
    finally:
        kernels.hash_seeds = orig
    assert sum(calls) > 20                                       # the device kernel really ran
